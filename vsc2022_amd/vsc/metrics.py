"""Records, CSV formats and the two challenge metrics (candidate micro-AP, segment-level AP).

Public names and semantics follow the reference's `vsc/metrics.py` (cited per symbol; paths relative
to /root/reference) so that files and scores are interchangeable; the implementation is array-first
(10^5..10^6 predictions must not spend minutes in per-object Python).  Values are pinned against the
reference by fixture `tests/golden/g7_metrics.npz`.
"""
import bisect
import dataclasses
import enum
import math
from typing import Collection, Dict, Iterable, List, NamedTuple, Optional, TextIO, Tuple, Union

import numpy as np


class Dataset(enum.Enum):
    """Which side of the challenge a video id belongs to; the value is the id prefix (metrics.py:21-23)."""

    QUERIES = "Q"
    REFS = "R"


def format_video_id(video_id: Union[str, int], dataset: Optional[Dataset]) -> str:
    """Canonical string id: integers become `<prefix>%06d`, strings are checked against the prefix
    (metrics.py:26-40)."""
    if isinstance(video_id, (int, np.integer)):
        if dataset is None:
            raise ValueError("Unable to convert integer video_id without a Dataset enum")
        return "%s%06d" % (dataset.value, int(video_id))
    if not isinstance(video_id, str):
        raise AssertionError(f"unexpected video_id: {video_id} of type {type(video_id)}")
    if dataset is not None and not video_id.startswith(dataset.value):
        raise AssertionError(f"dataset mismatch? got {video_id} for dataset {dataset}")
    return video_id


def _pandas():
    import pandas  # deferred: `import vsc2022_amd` stays light

    return pandas


def _pair_columns(pairs) -> Tuple[list, list, np.ndarray]:
    """(query ids, ref ids, scores) of a collection of CandidatePair (array-backed lists expose
    `columns()` and are never expanded into objects)."""
    columns = getattr(pairs, "columns", None)
    if columns is not None:
        q, r, s = columns()
        return list(q), list(r), np.asarray(s, dtype=np.float64)
    pairs = list(pairs)
    return ([p.query_id for p in pairs], [p.ref_id for p in pairs],
            np.fromiter((p.score for p in pairs), dtype=np.float64, count=len(pairs)))


@dataclasses.dataclass
class CandidatePair:
    """A (query video, reference video) pair with a confidence (metrics.py:43-93)."""

    query_id: str
    ref_id: str
    score: float

    @classmethod
    def to_dataframe(cls, candidates: Collection["CandidatePair"]):
        q, r, _ = _pair_columns(candidates)
        if not q:  # the reference builds a column-less frame from an empty list
            return _pandas().DataFrame([])
        # scores keep their own dtype (fp32 from the engine) so that the CSV text matches the reference's
        raw_scores = candidates.columns()[2] if hasattr(candidates, "columns") else [c.score for c in candidates]
        return _pandas().DataFrame({
            "query_id": [format_video_id(x, Dataset.QUERIES) for x in q],
            "ref_id": [format_video_id(x, Dataset.REFS) for x in r],
            "score": raw_scores,
        })

    @classmethod
    def write_csv(cls, candidates: Collection["CandidatePair"], file: Union[str, TextIO]):
        cls.to_dataframe(candidates).to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file: Union[str, TextIO]) -> List["CandidatePair"]:
        table = _pandas().read_csv(file)
        rows = zip(table["query_id"], table["ref_id"], table["score"])
        return [cls(format_video_id(q, Dataset.QUERIES), format_video_id(r, Dataset.REFS), s) for q, r, s in rows]

    @classmethod
    def from_matches(cls, matches: Collection["Match"]) -> List["CandidatePair"]:
        """One pair per (query, ref) with the best segment score, floored at 0 as the reference's
        defaultdict(float) does (metrics.py:84-93)."""
        best: Dict[Tuple[str, str], float] = {}
        for m in matches:
            key = (m.query_id, m.ref_id)
            best[key] = max(m.score, best.get(key, 0.0))
        return [cls(q, r, s) for (q, r), s in best.items()]


@dataclasses.dataclass
class PrecisionRecallCurve:
    """metrics.py:96-111"""

    precisions: np.ndarray
    recalls: np.ndarray
    scores: np.ndarray

    def plot(self, ax=None, **kwargs):
        if ax is None:
            import matplotlib.pyplot as plt

            ax = plt.subplots()[1]
            ax.set(xlabel="recall", ylabel="precision", xlim=(0, 1.05), ylim=(0, 1.05))
        ax.plot(self.recalls, self.precisions, **kwargs)
        return ax


@dataclasses.dataclass
class AveragePrecision:
    """metrics.py:114-118"""

    ap: float
    pr_curve: PrecisionRecallCurve
    simple_ap: Optional[float] = None


# ------------------------------------------------------------------------------------ intervals

def _coalesce(intervals: Iterable[Tuple[float, float]]) -> List[Tuple[float, float]]:
    """Union of closed intervals as a sorted list of disjoint ones (touching intervals merge)."""
    merged: List[Tuple[float, float]] = []
    for lo, hi in sorted(intervals):
        if merged and lo <= merged[-1][1]:
            if hi > merged[-1][1]:
                merged[-1] = (merged[-1][0], hi)
        else:
            merged.append((lo, hi))
    return merged


class Intervals:
    """A union of intervals on one time axis (metrics.py:120-174)."""

    intervals: List[Tuple[float, float]]

    def __init__(self, intervals: Optional[List[Tuple[float, float]]] = None):
        self.intervals = _coalesce(intervals or [])

    def add(self, interval: Tuple[float, float]):
        """Insert one interval, keeping the list sorted and disjoint (O(log n + merged))."""
        lo, hi = interval
        items = self.intervals
        k = bisect.bisect_left(items, (lo, hi))
        if k > 0 and items[k - 1][1] >= lo:  # overlaps its left neighbour
            k -= 1
            lo, hi = items[k][0], max(items[k][1], hi)
        end = k
        while end < len(items) and items[end][0] <= hi:
            hi = max(hi, items[end][1])
            end += 1
        items[k:end] = [(lo, hi)]

    def union(self, intervals: "Intervals") -> "Intervals":
        return Intervals(self.intervals + intervals.intervals)

    def total_length(self) -> float:
        total = 0.0
        for lo, hi in self.intervals:
            total += hi - lo
        return total

    def intersect_length(self, intervals: "Intervals") -> float:
        """|A n B| = |A| + |B| - |A u B|."""
        return self.total_length() + intervals.total_length() - self.union(intervals).total_length()

    def __repr__(self):
        return str(self.intervals)

    __str__ = __repr__


class Axis(enum.Enum):
    QUERY = enum.auto()
    REF = enum.auto()


class Match(NamedTuple):
    """A copied segment: ground truth or prediction (metrics.py:182-235)."""

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
        return (self.query_start, self.query_end) if axis == Axis.QUERY else (self.ref_start, self.ref_end)

    def intersection_area(self, bbox: "Match") -> float:
        width = min(self.query_end, bbox.query_end) - max(self.query_start, bbox.query_start)
        height = min(self.ref_end, bbox.ref_end) - max(self.ref_start, bbox.ref_start)
        return abs(max(width, 0) * max(height, 0))

    def overlaps(self, bbox: "Match") -> bool:
        return self.intersection_area(bbox) > 0.0

    @classmethod
    def write_csv(cls, matches: Collection["Match"], file: Union[str, TextIO]):
        _pandas().DataFrame([m._asdict() for m in matches], columns=cls._fields).to_csv(file, index=False)

    @classmethod
    def read_csv(cls, file: Union[str, TextIO], is_gt=False, check=True) -> List["Match"]:
        table = _pandas().read_csv(file)
        table["query_id"] = [format_video_id(x, Dataset.QUERIES) for x in table["query_id"]]
        table["ref_id"] = [format_video_id(x, Dataset.REFS) for x in table["ref_id"]]
        if is_gt:
            table["score"] = 1.0
        if check:
            for field in cls._fields:
                assert not table[field].isna().any()
        return [cls(**row) for row in table.to_dict("records")]


class VideoPair:
    """Running overlap statistics of one (query, ref) pair (metrics.py:238-301).

    A ground-truth segment counts towards the intersection only once some prediction overlaps it in
    area (both axes), as in the reference; the state is incremental -- which ground-truth segments
    are "awake" and the merged prediction intervals are carried from call to call instead of being
    rebuilt from every earlier prediction.
    """

    gts: List[Match]
    preds: List[Match]

    def __init__(self):
        self.gts, self.preds = [], []
        self.intersections = {axis: 0.0 for axis in Axis}
        self.totals = {axis: 0.0 for axis in Axis}
        self._awake: List[bool] = []
        self._covered = {axis: Intervals() for axis in Axis}

    def total_gt_length(self, axis: Axis) -> float:
        return Intervals([g.interval(axis) for g in self.gts]).total_length()

    def total_pred_length(self, axis: Axis) -> float:
        return Intervals([p.interval(axis) for p in self.preds]).total_length()

    def gt_overlaps(self, gt: Match) -> bool:
        return any(gt.overlaps(p) for p in self.preds)

    def add_gt(self, bbox: Match):
        self.gts.append(bbox)
        self._awake.append(self.gt_overlaps(bbox))

    def add_prediction(self, bbox: Match) -> Tuple[Dict[Axis, float], Dict[Axis, float]]:
        """Returns the change of (intersection with ground truth, covered length) per axis."""
        self.preds.append(bbox)
        for k, gt in enumerate(self.gts):
            if not self._awake[k] and gt.overlaps(bbox):
                self._awake[k] = True
        d_inter, d_total = {}, {}
        for axis in Axis:
            covered = self._covered[axis]
            covered.add(bbox.interval(axis))
            truth = Intervals([g.interval(axis) for g, on in zip(self.gts, self._awake) if on])
            inter, total = covered.intersect_length(truth), covered.total_length()
            d_inter[axis], d_total[axis] = inter - self.intersections[axis], total - self.totals[axis]
            self.intersections[axis], self.totals[axis] = inter, total
        return d_inter, d_total


def match_metric(gts: Collection[Match], predictions: Collection[Match]) -> AveragePrecision:
    """Segment-level AP of the matching track (metrics.py:304-378): AP = sum_i P(i) * dR(i) with
    P = sqrt(P_query * P_ref) and R = sqrt(R_query * R_ref); predictions with equal scores enter as
    one group before the curve is sampled."""
    ranked = sorted(predictions, key=lambda m: m.score, reverse=True)
    per_pair: Dict[Tuple[str, str], VideoPair] = {}
    for gt in gts:
        per_pair.setdefault(gt.pair_id(), VideoPair()).add_gt(gt)
    # per-axis ground-truth length, accumulated pair by pair (the reference's summation order)
    truth_length = {axis: 0.0 for axis in Axis}
    for pair in per_pair.values():
        for axis in Axis:
            truth_length[axis] += pair.total_gt_length(axis)

    inter = {axis: 0.0 for axis in Axis}
    covered = {axis: 0.0 for axis in Axis}
    ap, recall = 0.0, 0.0
    points = []  # (recall, precision, score)
    cursor = 0
    while cursor < len(ranked):
        score = ranked[cursor].score
        while cursor < len(ranked) and ranked[cursor].score == score:
            pred = ranked[cursor]
            d_inter, d_total = per_pair.setdefault(pred.pair_id(), VideoPair()).add_prediction(pred)
            for axis in Axis:
                inter[axis] += d_inter[axis]
                covered[axis] += d_total[axis]
            cursor += 1
        r_now = math.sqrt((inter[Axis.QUERY] / truth_length[Axis.QUERY]) * (inter[Axis.REF] / truth_length[Axis.REF]))
        p_now = math.sqrt((inter[Axis.QUERY] / covered[Axis.QUERY]) * (inter[Axis.REF] / covered[Axis.REF]))
        gain = r_now - recall
        ap += p_now * gain
        recall = r_now
        if gain > 0:
            points.append((recall, p_now, score))
    curve = PrecisionRecallCurve(np.array([p[1] for p in points]), np.array([p[0] for p in points]),
                                 np.array([p[2] for p in points]))
    return AveragePrecision(ap, curve)


@dataclasses.dataclass
class MatchingTrackMetrics:
    """metrics.py:381-386"""

    segment_ap: AveragePrecision      # the matching-track metric
    pairwise_micro_ap: AveragePrecision  # pair retrieval only, no localisation


def evaluate_matching_track(ground_truth_filename: str, predictions_filename: str) -> MatchingTrackMetrics:
    """Both metrics from two CSV files with the `Match` columns in any order (metrics.py:389-415)."""
    truth = Match.read_csv(ground_truth_filename, is_gt=True)
    predicted = Match.read_csv(predictions_filename)
    pair_ap = average_precision(CandidatePair.from_matches(truth), CandidatePair.from_matches(predicted))
    return MatchingTrackMetrics(segment_ap=match_metric(truth, predicted), pairwise_micro_ap=pair_ap)


def average_precision(ground_truth: Collection[CandidatePair], predictions: Collection[CandidatePair]
                      ) -> AveragePrecision:
    """Micro-AP over (query, ref) pairs (metrics.py:418-450): `ap` is the DrivenData value
    (threshold-grouped AP scaled by the share of the ground truth that was predicted at all),
    `simple_ap` the rank-based one."""
    gq, gr, _ = _pair_columns(ground_truth)
    wanted = set(zip(gq, gr))
    if len(wanted) != len(gq):
        raise AssertionError("Duplicates detected in ground truth")
    pq, pr, scores = _pair_columns(predictions)
    if len(set(zip(pq, pr))) != len(pq):
        raise AssertionError("Duplicates detected in predictions")
    canonical = drivendata_average_precision(predicted=CandidatePair.to_dataframe(predictions),
                                             ground_truth=CandidatePair.to_dataframe(ground_truth))
    order = np.argsort(-scores, kind="stable")
    hit = np.fromiter(((pq[k], pr[k]) in wanted for k in order), dtype=bool, count=len(order))
    found = np.cumsum(hit)
    precision = found / (np.arange(len(hit)) + 1)
    recall = found / len(wanted)
    simple_ap = np.sum(precision * hit) / len(wanted)
    at = np.flatnonzero(hit)
    return AveragePrecision(ap=canonical, pr_curve=PrecisionRecallCurve(precision[at], recall[at], scores[order][at]),
                            simple_ap=simple_ap)


def _threshold_ap(labels: np.ndarray, scores: np.ndarray) -> float:
    """Binary average precision as sklearn.metrics.average_precision_score defines it: over the
    distinct score thresholds, descending, sum (R_k - R_{k-1}) * P_k."""
    order = np.argsort(-scores, kind="stable")
    labels, scores = labels[order], scores[order]
    group_end = np.r_[np.flatnonzero(np.diff(scores)), len(scores) - 1]
    tp = np.cumsum(labels)[group_end]
    precision = tp / (group_end + 1)
    recall = tp / tp[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def drivendata_average_precision(predicted, ground_truth) -> float:
    """The challenge backend's AP (metrics.py:453-489); both arguments are frames with query_id /
    ref_id (/ score) columns."""
    if len(predicted) == 0:
        return 0.0
    scores = np.asarray(predicted["score"], dtype=np.float64)
    if not np.isfinite(scores).all():
        raise ValueError("Scores must be finite.")
    truth = set(zip(ground_truth["query_id"], ground_truth["ref_id"])) if len(ground_truth) else set()
    labels = np.fromiter((pair in truth for pair in zip(predicted["query_id"], predicted["ref_id"])),
                         dtype=np.float64, count=len(scores))
    n_found = int(labels.sum())
    if n_found == 0:
        return 0.0
    n_truth = int(_pandas().notna(ground_truth["ref_id"]).sum())
    return _threshold_ap(labels, scores) * (n_found / n_truth)
