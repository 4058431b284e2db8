"""Indexing step of the DnS baseline: run a student network over stored frame features and write the indexed
descriptors (optionally score-normalised) back as `.npz` files.

Drop-in for `python -m vsc.baseline.dns_index` of the reference (same flags, same output file names
`queries_<network>[_sn].npz` / `refs_<network>[_sn].npz`, same names `Accelerator`, `index_videos`, `main`;
/root/reference/vsc/baseline/dns_index.py:36-180).  The student is any TorchScript module with the attributes the
reference reads (`student_type` in {"cg", "fg"}, `fg_type`, `index_video`, `get_network_name`); PyTorch-ROCm runs it
("cuda" is the MI355X under ROCm), score normalisation goes through the engine's `score_normalize`.
"""
import argparse
import dataclasses
import enum
import logging
import os
from typing import List

import torch

from vsc2022_amd.vsc import metrics as M
from vsc2022_amd.vsc import storage
from vsc2022_amd.vsc.baseline.score_normalization import score_normalize
from vsc2022_amd.vsc.index import VideoFeature

logger = logging.getLogger("dns_index.py")
logger.setLevel(logging.INFO)


class Accelerator(enum.Enum):
    CPU = enum.auto()
    CUDA = enum.auto()

    def get_device(self) -> torch.device:
        return torch.device("cpu") if self is Accelerator.CPU else torch.device("cuda")


@torch.no_grad()
def index_videos(model, features: List[VideoFeature], device: torch.device) -> List[VideoFeature]:
    """Reference :101-118: coarse students see one region per frame; fine students keep region descriptors, binarised
    (`> 0`) or stored in half precision."""
    out = []
    for video in features:
        x = torch.from_numpy(video.feature).to(device)
        if model.student_type == "cg":
            x = x.unsqueeze(1)
        x = model.index_video(x.float())
        if model.student_type == "fg":
            x = x > 0 if model.fg_type == "bin" else x.half()
        out.append(dataclasses.replace(video, feature=x.cpu().numpy()))
    return out


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="DnS indexing on MI355X")
    p.add_argument("--query_features", help="Path to query descriptors", type=str, required=True)
    p.add_argument("--ref_features", help="Path to reference descriptors", type=str, required=True)
    p.add_argument("--score_norm_features", help="Path to score normalization descriptors", type=str)
    p.add_argument("--output_path", help="The path to write match predictions.", type=str, required=True)
    p.add_argument("--accelerator", choices=[x.name.lower() for x in Accelerator], default="cpu", type=str)
    p.add_argument("--torchscript_path", help="Path to student model used for indexing.", type=str, required=True)
    return p


parser = build_parser()


def main(args):
    model = torch.jit.load(args.torchscript_path)
    if "fg" in model.student_type and args.score_norm_features:
        raise Exception(f"Student type {model.student_type} can not be combined with score normalization.")
    device = Accelerator[args.accelerator.upper()].get_device()
    model = model.eval().to(device)
    extension = model.get_network_name()
    queries = index_videos(model, storage.load_features(args.query_features, M.Dataset.QUERIES), device)
    refs = index_videos(model, storage.load_features(args.ref_features, M.Dataset.REFS), device)
    logger.info("indexed %d queries and %d refs with %s", len(queries), len(refs), extension)
    if args.score_norm_features:
        noise = index_videos(model, storage.load_features(args.score_norm_features, M.Dataset.REFS), device)
        queries, refs = score_normalize(queries, refs, noise, replace_dim=False, beta=1.2)
        extension += "_sn"
    os.makedirs(args.output_path, exist_ok=True)
    storage.store_features(os.path.join(args.output_path, f"queries_{extension}.npz"), queries)
    storage.store_features(os.path.join(args.output_path, f"refs_{extension}.npz"), refs)


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    main(parser.parse_args())
