"""Descriptor files.

One `.npz` holds a whole dataset as three row-aligned arrays -- `video_ids` (one id string per
frame row), `features` [rows, dim] and `timestamps` [rows] or [rows, 2] -- exactly the container of
the reference (`vsc/storage.py:13-68`; pinned by fixture g3 and tests/test_storage.py of the
reference).  Videos are recovered from the runs of equal ids.
"""
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from vsc2022_amd.vsc.index import VideoFeature
from vsc2022_amd.vsc.metrics import Dataset, format_video_id


def store_features(f, features: List[VideoFeature], dataset: Optional[Dataset] = None) -> None:
    """Write `features` to the file (name or handle) `f`; integer ids get the dataset prefix."""
    per_video_ids = np.asarray([format_video_id(v.video_id, dataset) for v in features])
    rows_per_video = [len(v) for v in features]
    arrays = {
        "video_ids": np.repeat(per_video_ids, rows_per_video),
        "features": np.concatenate([v.feature for v in features]),
        "timestamps": np.concatenate([v.timestamps for v in features]),
    }
    np.savez(f, **arrays)


def same_value_ranges(values) -> Iterator[Tuple[object, int, int]]:
    """(value, start, end) for every maximal run of equal consecutive values."""
    values = np.asarray(values)
    if values.shape[0] == 0:
        raise IndexError("same_value_ranges of an empty sequence")
    change = np.flatnonzero(values[1:] != values[:-1]) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [values.shape[0]]])
    for a, b in zip(starts.tolist(), ends.tolist()):
        yield values[a], a, b


def _check_shapes(feats: np.ndarray, timestamps: np.ndarray) -> None:
    if timestamps.shape[0] != feats.shape[0]:
        raise ValueError(
            f"Expected the same number of timestamps as features: got "
            f"{timestamps.shape[0]} timestamps for {feats.shape[0]} features"
        )
    if timestamps.ndim != 1 and timestamps.shape[1:] != (2,):
        raise ValueError(f"Unexpected timestamp shape. Got {timestamps.shape}")


def load_features(f, dataset: Optional[Dataset] = None) -> List[VideoFeature]:
    """Read a descriptor file back into one VideoFeature per video (views into the file's arrays)."""
    with np.load(f, allow_pickle=False) as data:
        ids, feats, timestamps = data["video_ids"], data["features"], data["timestamps"]
    _check_shapes(feats, timestamps)
    videos = []
    for raw_id, a, b in same_value_ranges(ids):
        videos.append(VideoFeature(video_id=format_video_id(raw_id, dataset), timestamps=timestamps[a:b],
                                   feature=feats[a:b, :]))
    return videos


def convert_to_dict(features: List[VideoFeature]) -> Dict[str, VideoFeature]:
    return {v.video_id: v for v in features}
