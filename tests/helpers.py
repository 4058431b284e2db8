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


# ---- fixture g8 (BASELINE configs[0] shape): inputs are regenerated from the seed, the fixture holds their digest
G8 = dict(seed=80, n_query=50, n_ref=50, dim=512, q_frames=(20, 20), r_frames=(20, 20), planted_frac=0.2, noise=0.05,
          copy_len=(8, 20))
G8_NOISE = dict(seed=81, n_videos=30, frames=(20, 20))


def g8_inputs():
    """(queries, refs, noise, gts) exactly as oracle/gen_golden.py:g8_inputs builds them."""
    import hashlib

    from vsc2022_amd import synth

    q, r, gts = synth.make_dataset(**G8)
    noise = synth.make_videos(np.random.default_rng(G8_NOISE["seed"]), G8_NOISE["n_videos"], G8["dim"],
                              G8_NOISE["frames"], "R")
    for k, v in enumerate(noise):
        v.video_id = f"R{900000 + k:06d}"
    h = hashlib.sha256()
    for vids in (q, r, noise):
        for v in vids:
            h.update(str(v.video_id).encode())
            h.update(np.ascontiguousarray(v.feature, dtype=np.float32).tobytes())
            h.update(np.ascontiguousarray(v.timestamps, dtype=np.float32).tobytes())
    return q, r, noise, gts, h.hexdigest()


# ---- fixture g9 (DnS localisation): the stand-in for the fine-grained student, shared with oracle/gen_golden.py
def dns_standin(fg_type):
    """Chamfer-style region similarity: mean over the query regions of the best reference region.  `fg_type`
    as the reference reads it (vsc/baseline/dns_baseline.py:134-137): "att" = real-valued region descriptors,
    "bin" = binary ones (the caller rescales {0, 1} to {-1, +1}; the inner product is divided by the width)."""
    import torch

    class Chamfer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fg_type = fg_type
            self.student_type = "fg"

        def forward(self, a, b):
            s = torch.einsum("ird,jsd->ijrs", a, b)
            if "bin" in self.fg_type:
                s = s / a.shape[-1]
            return s.max(dim=3).values.mean(dim=2)

    return Chamfer()


G9 = dict(seed=90, n_query=6, n_ref=7, dim=64, q_frames=(12, 45), r_frames=(12, 50), planted_frac=1.0, noise=0.03,
          copy_len=(8, 30))


def g9_inputs(fg_type):
    """(coarse queries, coarse refs, fine queries, fine refs): the fine descriptors are R = 4 regions x 16 dims per
    frame, derived from the coarse ones so that planted copies show in both."""
    from vsc2022_amd import synth

    q, r, _ = synth.make_dataset(**G9)
    rng = np.random.default_rng(G9["seed"] + 1)
    proj = rng.standard_normal((4, G9["dim"], 16)).astype(np.float32)

    def fine(v):
        x = np.einsum("ld,rde->lre", v.feature, proj) + 0.05 * rng.standard_normal((len(v.feature), 4, 16)).astype(np.float32)
        if "bin" in fg_type:
            return x > 0
        return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)

    return q, r, [fine(v) for v in q], [fine(v) for v in r]


def pair_scores(orc, a, b, metric=None, chunk=256):
    """score(a[k], b[k]) for every k through the oracle's score matrix (its diagonal, in chunks): one oracle call per
    256 pairs instead of one per pair."""
    import numpy as np

    out = np.empty(len(a), dtype=np.float32)
    for k0 in range(0, len(a), chunk):
        m = orc.scores(a[k0:k0 + chunk], b[k0:k0 + chunk]) if metric is None else orc.scores(a[k0:k0 + chunk], b[k0:k0 + chunk], metric)
        out[k0:k0 + chunk] = np.diagonal(m)
    return out


def check_localisation_sample(orc, q_feats, q_off, r_feats, r_off, pair_q, pair_r, nbox, boxes, bscore, bias, n=2000,
                              seed=0, tn_kw=None):
    """A stratified sample of localised pairs recomputed by the CPU oracle: orc.pair_sims (the fp32 fma chains + bias,
    vsc/baseline/localization.py:75-80) -> orc.tn (Temporal Network) -> MaxSim score `sims[x1:x2, y1:y2].max() - bias`
    (localization.py:88-92, half-open slice); boxes and score BITS must equal the engine's.

    q_feats / r_feats: torch tensors in HBM (the rows the aligner sees); *_off: numpy video offsets; pair_q / pair_r /
    nbox / boxes / bscore: numpy arrays of the localised pairs in rank order.  Strata: the first and the last 10 % of the
    sample by rank, the rest split between pairs with and without boxes.  Returns (pairs checked, boxes checked)."""
    import torch

    tn_kw = dict(tn_max_step=5, min_length=4) if tn_kw is None else tn_kw
    rng = np.random.default_rng(seed)
    n_pairs = len(pair_q)
    n = min(n, n_pairs)
    edge = max(1, n // 10)
    pick = set(range(min(edge, n_pairs))) | set(range(max(0, n_pairs - edge), n_pairs))
    with_box, without = np.flatnonzero(nbox > 0), np.flatnonzero(nbox == 0)
    rest = max(0, n - len(pick))
    for pool, share in ((with_box, rest // 2), (without, rest - rest // 2)):
        if len(pool):
            pick |= set(rng.choice(pool, min(share, len(pool)), replace=False).tolist())
    if len(pick) < n:  # (the strata overlap, or one of them is smaller than its share: fill up from the rest)
        rest_pool = np.setdiff1d(np.arange(n_pairs), np.fromiter(pick, dtype=np.int64), assume_unique=False)
        pick |= set(rng.choice(rest_pool, n - len(pick), replace=False).tolist())
    pick = np.array(sorted(pick), dtype=np.int64)
    n_boxes = 0
    for k in pick:
        qv, rv = int(pair_q[k]), int(pair_r[k])
        a = q_feats[int(q_off[qv]) : int(q_off[qv + 1])].cpu().numpy()
        b = r_feats[int(r_off[rv]) : int(r_off[rv + 1])].cpu().numpy()
        sims = orc.pair_sims(a, b, bias)
        exp = orc.tn(sims, **tn_kw)
        got = boxes[k, : int(nbox[k])].tolist()
        assert got == [list(e) for e in exp], (int(k), qv, rv, got, exp)
        for b_, (x1, y1, x2, y2) in enumerate(exp):
            score = np.float32(sims[x1:x2, y1:y2].max() - np.float32(bias))
            assert score.view(np.uint32) == np.float32(bscore[k, b_]).view(np.uint32), (int(k), b_, float(score), float(bscore[k, b_]))
        n_boxes += len(exp)
    return len(pick), n_boxes
