"""Shared helpers for the parity tests: rebuild video lists from the golden fixture arrays."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def videos(fx, prefix, cls):
    """fixture arrays <prefix>_{ids,lens,feats,ts} -> list of VideoFeature-like objects of `cls`."""
    ids, lens = fx[prefix + "_ids"], fx[prefix + "_lens"]
    feats, ts = fx[prefix + "_feats"], fx[prefix + "_ts"]
    cuts = np.r_[0, np.cumsum(lens)]
    return [cls(video_id=str(ids[k]), timestamps=ts[cuts[k]:cuts[k + 1]], feature=feats[cuts[k]:cuts[k + 1]])
            for k in range(len(ids))]


def row_maps(fx, prefix):
    lens = fx[prefix + "_lens"]
    return np.repeat(np.arange(len(lens), dtype=np.int32), lens), np.r_[0, np.cumsum(lens)]


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def flatten_pairmatches(pms):
    """List[PairMatches] (reference-shaped) -> the arrays gen_golden.pack_pairmatches stores."""
    rows = [(m.query_timestamps[0], m.query_timestamps[1], m.ref_timestamps[0], m.ref_timestamps[1], m.score)
            for pm in pms for m in pm.matches]
    return (np.array([str(pm.query_id) for pm in pms]), np.array([str(pm.ref_id) for pm in pms]),
            np.array([len(pm.matches) for pm in pms], dtype=np.int64),
            np.array(rows, dtype=np.float64).reshape(-1, 5), np.array([r[4] for r in rows], dtype=np.float32))
