"""DnS matching baseline on the MI355X engine: coarse-descriptor search -> candidates -> Temporal-Network
localisation on a FINE-GRAINED similarity produced by a torch network.

Drop-in for `python -m vsc.baseline.dns_baseline` of the reference (same flags, same output files, same
module-level names `VCSLLocalizationDnS`, `search`, `localize_and_verify`, `match`, `main`;
/root/reference/vsc/baseline/dns_baseline.py:52-290).

`VCSLLocalizationDnS` (reference :108-163): the frame x frame matrix handed to the aligner comes from a
caller-supplied similarity network on FINE (region-level) descriptors, optionally symmetrised, mapped to
[0, 1] and combined with the coarse inner-product similarity (+ bias) by a geometric mean.  The reference gets
the network from a TorchScript file (`dns_fine_grained_student`); no weights exist in this environment, so the
network is whatever `torch.jit.load` / the caller supplies: any callable `sim_model(query [Lq, ...], ref [Lr, ...])
-> [Lq, Lr]` with an `fg_type` attribute.  The fine descriptors arrive as the reference's callers pass them --
`Dict[video_id, VideoFeature]` (`convert_to_dict`, reference :183-186, 260-264) -- or as a list.

Because `similarity()` is overridden, `localize_all` takes the reference's route of
`vsc2022_amd.vsc.baseline.localization.VCSLLocalization`: one matrix per pair (the coarse part computed on the
GPU by libvscmi), alignment of all of them in one `vsc_tn_forward_sim` call, MaxSim score per box.
"""
import argparse
import logging
import os
from typing import Dict, List, Mapping, Sequence, Tuple, Union

import numpy as np
import torch

from vsc2022_amd.vsc import metrics as M
from vsc2022_amd.vsc import storage
from vsc2022_amd.vsc.baseline.dns_index import Accelerator
from vsc2022_amd.vsc.baseline.localization import VCSLLocalizationMaxSim
from vsc2022_amd.vsc.candidates import CandidateGeneration, MaxScoreAggregation
from vsc2022_amd.vsc.index import VideoFeature
from vsc2022_amd.vsc.metrics import CandidatePair, Match

logger = logging.getLogger("dns_baseline.py")
logger.setLevel(logging.INFO)

FineFeatures = Union[Mapping[object, VideoFeature], Sequence[VideoFeature]]


def _by_id(features: FineFeatures) -> Dict[object, VideoFeature]:
    if isinstance(features, Mapping):
        return dict(features)
    return {v.video_id: v for v in features}


class VCSLLocalizationDnS(VCSLLocalizationMaxSim):
    def __init__(self, model, queries_fine: FineFeatures, refs_fine: FineFeatures,
                 queries_coarse: List[VideoFeature], refs_coarse: List[VideoFeature], model_type, device,
                 symmetric: bool = True, geometric_mean: bool = True, **kwargs):
        super().__init__(queries_coarse, refs_coarse, model_type, **kwargs)
        self.queries_fine = _by_id(queries_fine)
        self.refs_fine = _by_id(refs_fine)
        self.sim_model = model
        # the reference stores what it was given (a torch.device or the --accelerator string) and only ever hands it
        # to Tensor.to(); "cuda" is the ROCm device under PyTorch-ROCm
        self.torch_device = device if isinstance(device, torch.device) else torch.device(device)
        self.symmetric = symmetric
        self.geometric_mean = geometric_mean

    def _rescale_binaries(self, x):
        if "bin" in getattr(self.sim_model, "fg_type", ""):
            x = 2 * x - 1
        return x

    @torch.no_grad()
    def similarity(self, candidate: CandidatePair):
        query = torch.from_numpy(np.asarray(self.queries_fine[candidate.query_id].feature)).to(self.torch_device).float()
        ref = torch.from_numpy(np.asarray(self.refs_fine[candidate.ref_id].feature)).to(self.torch_device).float()
        query, ref = self._rescale_binaries(query), self._rescale_binaries(ref)
        sim = self.sim_model(query, ref)
        if self.symmetric:
            sim = (sim + self.sim_model(ref, query).mT) / 2.0
        sim = (sim / 2.0 + 0.5).cpu().numpy()
        if self.geometric_mean:
            coarse = self._device_similarity(candidate, self.similarity_bias)  # q @ r.T + bias, on the GPU
            sim = np.sqrt(sim.clip(1e-7) * coarse.clip(1e-7))
        return sim


def search(queries: List[VideoFeature], refs: List[VideoFeature], retrieve_per_query: float = 1200.0,
           candidates_per_query: float = 25.0) -> List[CandidatePair]:
    """Best `candidates_per_query * len(queries)` video pairs on the COARSE descriptors (reference :166-180)."""
    n_hits = int(retrieve_per_query * len(queries))
    n_keep = int(candidates_per_query * len(queries))
    pairs = CandidateGeneration(refs, MaxScoreAggregation()).query(queries, global_k=n_hits)[:n_keep]
    logger.info("Got %d candidates", len(pairs))
    return pairs


def localize_and_verify(model, queries_fine: FineFeatures, refs_fine: FineFeatures, queries_coarse: List[VideoFeature],
                        refs_coarse: List[VideoFeature], candidates: Sequence[CandidatePair],
                        localize_per_query: float = 5.0, device="cpu") -> List[Match]:
    """Reference :183-227: the best `localize_per_query * len(queries_fine)` candidates, batches of 512 pairs."""
    todo = candidates[: int(len(queries_fine) * localize_per_query)]
    aligner = VCSLLocalizationDnS(model, queries_fine, refs_fine, queries_coarse, refs_coarse, model_type="TN",
                                  tn_max_step=5, min_length=4, concurrency=16, similarity_bias=0.5, device=device)
    found: List[Match] = []
    batch = 512
    for start in range(0, len(todo), batch):
        found += aligner.localize_all(todo[start : start + batch])
        logger.info("Aligned %d pairs of %d; %d predictions so far", min(start + batch, len(todo)), len(todo), len(found))
    return found


def match(model, queries_fine: FineFeatures, refs_fine: FineFeatures, queries_coarse: List[VideoFeature],
          refs_coarse: List[VideoFeature], output_path: str, device) -> Tuple[str, str]:
    """Reference :230-258; returns the paths of candidates.csv and matches.csv."""
    pairs = search(queries_coarse, refs_coarse)
    os.makedirs(output_path, exist_ok=True)
    candidate_file = os.path.join(output_path, "candidates.csv")
    CandidatePair.write_csv(pairs, candidate_file)
    found = localize_and_verify(model, queries_fine, refs_fine, queries_coarse, refs_coarse, pairs, device=device)
    matches_file = os.path.join(output_path, "matches.csv")
    Match.write_csv(found, matches_file)
    return candidate_file, matches_file


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="DnS matching baseline on MI355X")
    for flag, text in (("--torchscript_path", "Path to the fine-grained student model used for similarity calculation."),
                       ("--query_coarse_features", "Path to query coarse descriptors"),
                       ("--ref_coarse_features", "Path to reference coarse descriptors"),
                       ("--query_fine_features", "Path to query fine descriptors"),
                       ("--ref_fine_features", "Path to reference fine descriptors"),
                       ("--output_path", "The path to write match prediction.")):
        p.add_argument(flag, help=text, type=str, required=True)
    p.add_argument("--accelerator", help="Device used for the similarity calculation",
                   choices=[x.name.lower() for x in Accelerator], default="cpu", type=str)
    p.add_argument("--ground_truth", help="Path to the ground truth (labels) CSV file.", type=str)
    p.add_argument("--overwrite", help="Overwrite prediction files, if found.", action="store_true")
    return p


parser = build_parser()


def main(args):
    if os.path.exists(args.output_path) and not args.overwrite:
        raise Exception(f"Output path already exists: {args.output_path}. Do you want to --overwrite?")
    model = torch.jit.load(args.torchscript_path)
    if "fg" != model.student_type:
        raise Exception("Only fine-grained student are accepted for similarity calculation.")
    device = Accelerator[args.accelerator.upper()].get_device()
    model = model.eval().to(device)
    queries_fine = storage.convert_to_dict(storage.load_features(args.query_fine_features, M.Dataset.QUERIES))
    refs_fine = storage.convert_to_dict(storage.load_features(args.ref_fine_features, M.Dataset.REFS))
    queries_coarse = storage.load_features(args.query_coarse_features, M.Dataset.QUERIES)
    refs_coarse = storage.load_features(args.ref_coarse_features, M.Dataset.REFS)
    candidate_file, match_file = match(model, queries_fine, refs_fine, queries_coarse, refs_coarse, args.output_path,
                                       device)
    if args.ground_truth:
        from vsc2022_amd.vsc.baseline.sscd_baseline import _report

        _report(args.output_path, args.ground_truth, candidate_file, match_file)


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    main(parser.parse_args())
