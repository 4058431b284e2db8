"""Record types, CSV formats and evaluation metrics of the matching pipeline.

Host-side mirror of the reference's `vsc/metrics.py` (same public names and semantics, cited per
symbol below; paths relative to /root/reference).  Nothing here is GPU work: these are the
O(n log n) judges (micro-AP, segment AP) that run once per evaluation.  They are written
array-first so that 10^5..10^6 predictions do not spend minutes in Python loops.
"""
import collections
import dataclasses
import enum
import math
from typing import Collection, Dict, Iterable, List, NamedTuple, Optional, Sequence, TextIO, Tuple, Union

import numpy as np


class Dataset(enum.Enum):
    """vsc/metrics.py:21-23"""

    QUERIES = "Q"
    REFS = "R"


def format_video_id(video_id: Union[str, int], dataset: Optional[Dataset]) -> str:
    """vsc/metrics.py:26-40: ints become e.g. Q000123; strings are checked against the dataset."""
    if isinstance(video_id, (int, np.integer)):
        if dataset is None:
            raise ValueError("Unable to convert integer video_id without a Dataset enum")
        return "%s%06d" % (dataset.value, int(video_id))
    if not isinstance(video_id, str):
        raise AssertionError(f"unexpected video_id: {video_id} of type {type(video_id)}")
    if dataset is not None and video_id[0] != dataset.value:
        raise AssertionError(f"dataset mismatch? got {video_id} for dataset {dataset}")
    return video_id


def _pd():
    import pandas as pd  # deferred: keeps `import vsc2022_amd` light

    return pd


@dataclasses.dataclass
class CandidatePair:
    """vsc/metrics.py:43-93"""

    query_id: str
    ref_id: str
    score: float

    @classmethod
    def to_dataframe(cls, candidates: Collection["CandidatePair"]):
        cols = getattr(candidates, "columns", None)
        if cols is not None:  # array-backed CandidateList: no per-object work
            qids, rids, scores = cols()
        else:
            qids = [c.query_id for c in candidates]
            rids = [c.ref_id for c in candidates]
            scores = [c.score for c in candidates]
        frame = _pd().DataFrame(
            {
                "query_id": [format_video_id(q, Dataset.QUERIES) for q in qids],
                "ref_id": [format_video_id(r, Dataset.REFS) for r in rids],
                "score": scores,
            }
        )
        if len(frame) == 0:  # the reference builds an empty, column-less frame from []
            frame = _pd().DataFrame([])
        return frame

    @classmethod
    def write_csv(cls, candidates: Collection["CandidatePair"], file: Union[str, TextIO]):
        cls.to_dataframe(candidates).to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file: Union[str, TextIO]) -> List["CandidatePair"]:
        frame = _pd().read_csv(file)
        return [
            CandidatePair(
                query_id=format_video_id(q, Dataset.QUERIES),
                ref_id=format_video_id(r, Dataset.REFS),
                score=s,
            )
            for q, r, s in zip(frame["query_id"], frame["ref_id"], frame["score"])
        ]

    @classmethod
    def from_matches(cls, matches: Collection["Match"]) -> List["CandidatePair"]:
        """Best score per (query, ref); starts from 0.0 like defaultdict(float) (metrics.py:84-93)."""
        best: Dict[Tuple[str, str], float] = {}
        for m in matches:
            key = (m.query_id, m.ref_id)
            best[key] = max(m.score, best.get(key, 0.0))
        return [CandidatePair(query_id=q, ref_id=r, score=s) for (q, r), s in best.items()]


@dataclasses.dataclass
class PrecisionRecallCurve:
    """vsc/metrics.py:96-111"""

    precisions: np.ndarray
    recalls: np.ndarray
    scores: np.ndarray

    def plot(self, ax=None, **kwargs):
        if ax is None:
            import matplotlib.pyplot as plt

            _, ax = plt.subplots()
            ax.set_xlabel("recall")
            ax.set_ylabel("precision")
            ax.set_xlim(0, 1.05)
            ax.set_ylim(0, 1.05)
        ax.plot(self.recalls, self.precisions, **kwargs)
        return ax


@dataclasses.dataclass
class AveragePrecision:
    """vsc/metrics.py:114-118"""

    ap: float
    pr_curve: PrecisionRecallCurve
    simple_ap: Optional[float] = None


def _merge_intervals(intervals: Iterable[Tuple[float, float]]) -> List[Tuple[float, float]]:
    out: List[Tuple[float, float]] = []
    for start, end in sorted(intervals):
        if out and start <= out[-1][1]:
            if end > out[-1][1]:
                out[-1] = (out[-1][0], end)
        else:
            out.append((start, end))
    return out


class Intervals:
    """Set of non-overlapping intervals ordered by start (vsc/metrics.py:120-174)."""

    intervals: List[Tuple[float, float]]

    def __init__(self, intervals: Optional[List[Tuple[float, float]]] = None):
        self.intervals = _merge_intervals(intervals or [])

    def add(self, interval: Tuple[float, float]):
        self.intervals = _merge_intervals(self.intervals + [interval])

    def union(self, intervals: "Intervals") -> "Intervals":
        return Intervals(self.intervals + intervals.intervals)

    def total_length(self) -> float:
        length = 0.0
        for start, end in self.intervals:
            length += end - start
        return length

    def intersect_length(self, intervals: "Intervals") -> float:
        """|A n B| = |A| + |B| - |A U B|"""
        return self.total_length() + intervals.total_length() - self.union(intervals).total_length()

    def __str__(self):
        return str(self.intervals)

    __repr__ = __str__


class Axis(enum.Enum):
    QUERY = enum.auto()
    REF = enum.auto()


class Match(NamedTuple):
    """A ground-truth or predicted copied segment (vsc/metrics.py:182-235)."""

    query_id: str
    ref_id: str
    score: float
    query_start: float
    query_end: float
    ref_start: float
    ref_end: float

    def pair_id(self):
        return (self.query_id, self.ref_id)

    def interval(self, axis: Axis) -> Tuple[float, float]:
        if axis == Axis.QUERY:
            return (self.query_start, self.query_end)
        return (self.ref_start, self.ref_end)

    def intersection_area(self, bbox: "Match") -> float:
        dq = min(self.query_end, bbox.query_end) - max(self.query_start, bbox.query_start)
        dr = min(self.ref_end, bbox.ref_end) - max(self.ref_start, bbox.ref_start)
        return abs(max(dq, 0) * max(dr, 0))

    def overlaps(self, bbox: "Match") -> bool:
        return self.intersection_area(bbox) > 0.0

    @classmethod
    def write_csv(cls, matches: Collection["Match"], file: Union[str, TextIO]):
        frame = _pd().DataFrame([m._asdict() for m in matches], columns=cls._fields)
        frame.to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file: Union[str, TextIO], is_gt=False, check=True) -> List["Match"]:
        frame = _pd().read_csv(file)
        frame["query_id"] = frame.query_id.map(lambda x: format_video_id(x, Dataset.QUERIES))
        frame["ref_id"] = frame.ref_id.map(lambda x: format_video_id(x, Dataset.REFS))
        if is_gt:
            frame["score"] = 1.0
        if check:
            for field in cls._fields:
                assert not frame[field].isna().any()
        return [Match(**record) for record in frame.to_dict("records")]


class VideoPair:
    """Per (query, ref) accumulator of predictions against ground truth (vsc/metrics.py:238-301).

    Ground-truth boxes only count towards the intersection once at least one prediction overlaps
    them (area overlap), exactly as the reference; the bookkeeping is incremental: the set of
    overlapped GT boxes and the merged prediction intervals are carried between calls instead of
    being rebuilt from every past prediction.
    """

    gts: List[Match]
    preds: List[Match]

    def __init__(self):
        self.intersections = {axis: 0.0 for axis in Axis}
        self.totals = {axis: 0.0 for axis in Axis}
        self.gts = []
        self.preds = []
        self._gt_hit: List[bool] = []
        self._pred_ints = {axis: Intervals() for axis in Axis}

    def total_gt_length(self, axis: Axis) -> float:
        return Intervals([gt.interval(axis) for gt in self.gts]).total_length()

    def total_pred_length(self, axis: Axis) -> float:
        return Intervals([pred.interval(axis) for pred in self.preds]).total_length()

    def gt_overlaps(self, gt: Match) -> bool:
        return any(gt.overlaps(pred) for pred in self.preds)

    def add_gt(self, bbox: Match):
        self.gts.append(bbox)
        self._gt_hit.append(self.gt_overlaps(bbox))

    def add_prediction(self, bbox: Match) -> Tuple[Dict[Axis, float], Dict[Axis, float]]:
        self.preds.append(bbox)
        for k, gt in enumerate(self.gts):
            if not self._gt_hit[k] and gt.overlaps(bbox):
                self._gt_hit[k] = True
        considered = [gt for gt, hit in zip(self.gts, self._gt_hit) if hit]
        intersect_deltas, total_deltas = {}, {}
        for axis in Axis:
            self._pred_ints[axis].add(bbox.interval(axis))
            pred_ints = self._pred_ints[axis]
            gt_ints = Intervals([gt.interval(axis) for gt in considered])
            intersect_length = pred_ints.intersect_length(gt_ints)
            prediction_length = pred_ints.total_length()
            intersect_deltas[axis] = intersect_length - self.intersections[axis]
            total_deltas[axis] = prediction_length - self.totals[axis]
            self.intersections[axis] = intersect_length
            self.totals[axis] = prediction_length
        return intersect_deltas, total_deltas


def match_metric(gts: Collection[Match], predictions: Collection[Match]) -> AveragePrecision:
    """Segment-level AP of the matching track (vsc/metrics.py:304-378).

    AP = sum_i P(i) dR(i) with P = sqrt(P_query * P_ref), R = sqrt(R_query * R_ref); predictions
    sharing a score are applied as one group before the curve is sampled.
    """
    ordered = sorted(predictions, key=lambda m: m.score, reverse=True)
    pairs: Dict[Tuple[str, str], VideoPair] = collections.defaultdict(VideoPair)
    for gt in gts:
        pairs[gt.pair_id()].add_gt(gt)
    gt_total = {axis: 0.0 for axis in Axis}
    for pair in pairs.values():
        for axis in Axis:
            gt_total[axis] += pair.total_gt_length(axis)

    inter = {axis: 0.0 for axis in Axis}
    total = {axis: 0.0 for axis in Axis}
    recall = 0.0
    metric = 0.0
    curve_r, curve_p, curve_s = [], [], []
    pos, n = 0, len(ordered)
    while pos < n:
        score = ordered[pos].score
        while pos < n and ordered[pos].score == score:
            d_inter, d_total = pairs[ordered[pos].pair_id()].add_prediction(ordered[pos])
            for axis in Axis:
                inter[axis] += d_inter[axis]
                total[axis] += d_total[axis]
            pos += 1
        rec = {axis: inter[axis] / gt_total[axis] for axis in Axis}
        prec = {axis: inter[axis] / total[axis] for axis in Axis}
        new_recall = math.sqrt(rec[Axis.QUERY] * rec[Axis.REF])
        precision = math.sqrt(prec[Axis.QUERY] * prec[Axis.REF])
        delta = new_recall - recall
        metric += precision * delta
        recall = new_recall
        if delta > 0:
            curve_r.append(recall)
            curve_p.append(precision)
            curve_s.append(score)
    return AveragePrecision(
        metric, PrecisionRecallCurve(np.array(curve_p), np.array(curve_r), np.array(curve_s))
    )


@dataclasses.dataclass
class MatchingTrackMetrics:
    """vsc/metrics.py:381-386"""

    segment_ap: AveragePrecision
    pairwise_micro_ap: AveragePrecision


def evaluate_matching_track(ground_truth_filename: str, predictions_filename: str) -> MatchingTrackMetrics:
    """vsc/metrics.py:389-415"""
    gt = Match.read_csv(ground_truth_filename, is_gt=True)
    predictions = Match.read_csv(predictions_filename)
    metric = match_metric(gt, predictions)
    pair_ap = average_precision(CandidatePair.from_matches(gt), CandidatePair.from_matches(predictions))
    return MatchingTrackMetrics(segment_ap=metric, pairwise_micro_ap=pair_ap)


def _columns(pairs: Collection[CandidatePair]):
    cols = getattr(pairs, "columns", None)
    if cols is not None:
        q, r, s = cols()
        return list(q), list(r), np.asarray(s, dtype=np.float64)
    return ([p.query_id for p in pairs], [p.ref_id for p in pairs],
            np.asarray([p.score for p in pairs], dtype=np.float64))


def average_precision(
    ground_truth: Collection[CandidatePair], predictions: Collection[CandidatePair]
) -> AveragePrecision:
    """Micro-AP over (query, ref) pairs (vsc/metrics.py:418-450).

    `ap` is the DrivenData-style value (threshold-grouped AP scaled by the fraction of ground
    truth that was predicted at all), `simple_ap` the rank-based one.
    """
    gq, gr, _ = _columns(ground_truth)
    gt_pairs = set(zip(gq, gr))
    if len(gt_pairs) != len(gq):
        raise AssertionError("Duplicates detected in ground truth")
    pq, pr, scores = _columns(predictions)
    if len(set(zip(pq, pr))) != len(pq):
        raise AssertionError("Duplicates detected in predictions")
    canonical = drivendata_average_precision(
        predicted=CandidatePair.to_dataframe(predictions),
        ground_truth=CandidatePair.to_dataframe(ground_truth),
    )
    order = np.argsort(-scores, kind="stable")
    scores = scores[order]
    correct = np.fromiter(((pq[k], pr[k]) in gt_pairs for k in order), dtype=bool, count=len(order))
    total_pairs = len(gt_pairs)
    cum_correct = np.cumsum(correct)
    recall = cum_correct / total_pairs
    precision = cum_correct / (np.arange(len(correct)) + 1)
    simple_ap = np.sum(precision * correct) / total_pairs
    hit = np.nonzero(correct)[0]
    curve = PrecisionRecallCurve(precision[hit], recall[hit], scores[hit])
    return AveragePrecision(ap=canonical, pr_curve=curve, simple_ap=simple_ap)


def _threshold_ap(labels: np.ndarray, scores: np.ndarray) -> float:
    """sklearn.metrics.average_precision_score for binary labels: sum over distinct score
    thresholds (descending) of (R_k - R_{k-1}) * P_k."""
    order = np.argsort(-scores, kind="stable")
    labels = labels[order]
    scores = scores[order]
    last_of_group = np.r_[np.nonzero(np.diff(scores))[0], len(scores) - 1]
    tp = np.cumsum(labels)[last_of_group]
    seen = last_of_group + 1
    precision = tp / seen
    recall = tp / tp[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def drivendata_average_precision(predicted, ground_truth) -> float:
    """Canonical challenge AP (vsc/metrics.py:453-489).  Frames carry query_id / ref_id / score."""
    if len(predicted) == 0:
        return 0.0
    scores = np.asarray(predicted["score"], dtype=np.float64)
    if not np.isfinite(scores).all():
        raise ValueError("Scores must be finite.")
    actual = set(zip(ground_truth["query_id"], ground_truth["ref_id"])) if len(ground_truth) else set()
    labels = np.fromiter(
        (pair in actual for pair in zip(predicted["query_id"], predicted["ref_id"])),
        dtype=np.float64,
        count=len(scores),
    )
    predicted_n_pos = int(labels.sum())
    if predicted_n_pos == 0:
        return 0.0
    actual_n_pos = int(_pd().notna(ground_truth["ref_id"]).sum())
    return _threshold_ap(labels, scores) * (predicted_n_pos / actual_n_pos)
