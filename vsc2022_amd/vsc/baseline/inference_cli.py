"""Descriptor inference from the command line, on PyTorch-ROCm (BASELINE configs[2]).

Accepts the flags of the reference's `python -m vsc.baseline.inference` (/root/reference/vsc/baseline/
inference.py:52-82) and mirrors its flow (`inference.py:93-158`, `inference_impl.py:169-247`):

    python -m vsc2022_amd.vsc.baseline.inference_cli --torchscript_path sscd_disc_mixup.no_l2_norm.torchscript.pt \
        --accelerator cuda --processes 8 --dataset_path videos/ --output_file out/queries.npz

  * videos of `--dataset_path` with one of `--video_extensions`, sorted, video i goes to the rank with
    `i % world == rank` (`inference_impl.py:100-109`);
  * frames at `--fps`, transformed on the DEVICE (`device_transform`: the three `InferenceTransforms`
    of `inference_impl.py:39-69` -- antialiased bilinear resize, centre crop, ToTensor, Normalize -- as
    tensor ops instead of PIL + torchvision on the host);
  * batches of <= `--batch_size` frames of ONE video through the TorchScript model (`torch.jit.load`,
    `inference_impl.py:173`) -> one VideoFeature per video with `(i / fps, (i + 1) / fps)` timestamps;
  * `--store_fp16`; one `.npz` per rank under `--scratch_path`, merged into `--output_file` by
    `merge_feature_files` (`inference_impl.py:242-247`); `--processes N` spawns N workers on this machine
    (one per device with `--accelerator cuda`), `--distributed_rank/--distributed_size` for externally
    launched ranks (then `--output_file` is this rank's file, as in the reference).

Deliberate differences: (1) `--video_reader NPY` reads `<name>.npy` uint8 stacks [frames, H, W, 3] sampled at
`--fps` -- there is no ffmpeg binary in the build environment; `FFMPEG` shells out to `--ffmpeg_path` exactly
like the reference's `video_reader/ffmpeg_video_reader.py:29-56` and fails loudly when the binary is missing;
(2) without `--torchscript_path` a random-init network of the SSCD architecture is used (benchmarks only: there
are no weights to download here) and the run says so; (3) DNS / DINO baselines are not provided.
"""
import argparse
import enum
import glob
import logging
import os
import shutil
import subprocess
import tempfile
from typing import Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from vsc2022_amd.vsc.baseline.inference import IMAGENET_MEAN, IMAGENET_STD, build_sscd_model
from vsc2022_amd.vsc.index import VideoFeature
from vsc2022_amd.vsc.storage import load_features, store_features

logger = logging.getLogger("inference.py")
logger.setLevel(logging.INFO)


class InferenceTransforms(enum.Enum):
    RESIZE_288 = enum.auto()          # aspect-ratio preserving resize of the short edge to 288
    RESIZE_320_CENTER = enum.auto()   # short edge to 320, then the centre 320 x 320 crop
    RESIZE_224_SQUARE = enum.auto()   # resize to 224 x 224


class Accelerator(enum.Enum):
    CPU = enum.auto()
    CUDA = enum.auto()


class VideoReaderType(enum.Enum):
    FFMPEG = enum.auto()
    NPY = enum.auto()


class Baseline(enum.Enum):
    SSCD = enum.auto()


# ------------------------------------------------------------------------------------------- transforms
def _short_edge_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision.transforms.Resize(int): the short edge becomes `size`, the long edge int(size * long / short)."""
    if h <= w:
        return size, int(size * w / h)
    return int(size * h / w), size


def device_transform(frames_u8: torch.Tensor, transform: InferenceTransforms) -> torch.Tensor:
    """uint8 frames [n, 3, H, W] -> normalised float32 [n, 3, h, w] on the frames' device.

    Resize = antialiased bilinear on the 0..255 values, rounded back to whole levels like a PIL image would be
    (`transforms.Resize` on the PIL frames the reference decodes), then ToTensor (/255) and Normalize."""
    if frames_u8.ndim == 3:
        frames_u8 = frames_u8.unsqueeze(0)
    n, c, h, w = frames_u8.shape
    x = frames_u8.float()
    if transform == InferenceTransforms.RESIZE_224_SQUARE:
        target = (224, 224)
    else:
        target = _short_edge_size(h, w, 288 if transform == InferenceTransforms.RESIZE_288 else 320)
    if target != (h, w):
        x = F.interpolate(x, size=target, mode="bilinear", antialias=True, align_corners=False)
        x = x.round_().clamp_(0.0, 255.0)
    if transform == InferenceTransforms.RESIZE_320_CENTER:
        th, tw = x.shape[2], x.shape[3]
        top, left = int(round((th - 320) / 2.0)), int(round((tw - 320) / 2.0))  # torchvision center_crop
        x = x[:, :, top : top + 320, left : left + 320]
    x = x / 255.0
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


# ------------------------------------------------------------------------------------------- readers
class NpyVideoReader:
    """`<name>.npy`: uint8 [frames, H, W, 3], already sampled at the required fps."""

    def __init__(self, video_path: str, required_fps: float):
        self.video_path, self.required_fps = video_path, float(required_fps)

    def frames(self) -> Iterator[Tuple[float, float, np.ndarray]]:
        stack = np.load(self.video_path, mmap_mode="r")
        if stack.ndim != 4 or stack.shape[3] != 3 or stack.dtype != np.uint8:
            raise ValueError(f"{self.video_path}: expected uint8 [frames, H, W, 3], got {stack.dtype} {stack.shape}")
        for i in range(stack.shape[0]):
            yield i / self.required_fps, (i + 1) / self.required_fps, np.asarray(stack[i])


class FFMpegVideoReader:
    """Frames at `required_fps` through the ffmpeg binary (video_reader/ffmpeg_video_reader.py:29-56)."""

    def __init__(self, video_path: str, required_fps: float, ffmpeg_path: str = "ffmpeg"):
        self.video_path, self.required_fps, self.ffmpeg_path = video_path, float(required_fps), ffmpeg_path
        if shutil.which(ffmpeg_path) is None:
            raise FileNotFoundError(f"ffmpeg binary '{ffmpeg_path}' not found (--ffmpeg_path); "
                                    "use --video_reader NPY for pre-decoded frame stacks")

    def frames(self) -> Iterator[Tuple[float, float, np.ndarray]]:
        from PIL import Image

        with tempfile.TemporaryDirectory() as tmp, open(os.devnull, "w") as null:
            subprocess.check_call([self.ffmpeg_path, "-nostdin", "-y", "-i", self.video_path, "-start_number", "0",
                                   "-q", "0", "-vf", "fps=%f" % self.required_fps, os.path.join(tmp, "%07d.png")],
                                  stderr=null)
            i = 0
            while os.path.exists(os.path.join(tmp, f"{i:07d}.png")):
                with Image.open(os.path.join(tmp, f"{i:07d}.png")) as img:
                    frame = np.asarray(img.convert("RGB"))
                yield i / self.required_fps, (i + 1) / self.required_fps, frame
                i += 1


class VideoDataset:
    """The videos of one rank, each as (name, timestamps [n, 2] float64, frames uint8 [n, H, W, 3])."""

    def __init__(self, path: str, fps: float, extensions=("mp4",), distributed_rank: int = 0,
                 distributed_world_size: int = 1, video_reader: VideoReaderType = VideoReaderType.FFMPEG,
                 ffmpeg_path: str = "ffmpeg"):
        assert distributed_rank < distributed_world_size
        names = []
        for ext in extensions:
            names.extend(glob.glob(os.path.join(path, f"*.{ext}")))
        self.videos = sorted(set(names))
        if not self.videos:
            raise Exception("No videos found!")
        self.fps, self.video_reader, self.ffmpeg_path = fps, video_reader, ffmpeg_path
        self.selected_videos = [(i, v) for i, v in enumerate(self.videos)
                                if i % distributed_world_size == distributed_rank]

    def num_videos(self) -> int:
        return len(self.selected_videos)

    def __iter__(self):
        for _, video in self.selected_videos:
            name = os.path.basename(video).split(".")[0]
            if self.video_reader == VideoReaderType.NPY:
                reader = NpyVideoReader(video, self.fps)
            else:
                reader = FFMpegVideoReader(video, self.fps, self.ffmpeg_path)
            ts, frames = [], []
            for t0, t1, frame in reader.frames():
                ts.append((t0, t1))
                frames.append(frame)
            if frames:
                yield name, np.array(ts, dtype=np.float64), np.stack(frames)


# ------------------------------------------------------------------------------------------- inference
@torch.no_grad()
def run_inference(dataset: Iterable, model, device, transform: InferenceTransforms, batch_size: int = 32,
                  store_fp16: bool = False) -> Iterator[VideoFeature]:
    """One VideoFeature per video; batches never mix videos (inference_impl.py:210-239)."""
    for name, ts, frames in dataset:
        outs = []
        for b0 in range(0, len(frames), batch_size):
            x = torch.from_numpy(np.ascontiguousarray(frames[b0 : b0 + batch_size])).to(device).permute(0, 3, 1, 2)
            y = model(device_transform(x, transform)).float().cpu()
            outs.append(y.half().numpy() if store_fp16 else y.numpy())
        yield VideoFeature(video_id=name, timestamps=ts, feature=np.concatenate(outs, axis=0))


def merge_feature_files(filenames: List[str], output_filename: str) -> int:
    features = []
    for fn in filenames:
        features.extend(load_features(fn))
    store_features(output_filename, features)
    return len(features)


def get_device(args, rank: int, world_size: int) -> torch.device:
    if Accelerator[args.accelerator.upper()] != Accelerator.CUDA:
        return torch.device("cpu")
    assert torch.cuda.is_available(), "--accelerator cuda needs a GPU"
    n = torch.cuda.device_count()
    if args.processes > n:
        raise Exception(f"Asked for {args.processes} processes and cuda, but only {n} devices found")
    dev = rank if (args.processes > 1 or world_size <= n) else 0
    torch.cuda.set_device(dev)
    return torch.device("cuda", dev)


class CheckedFast(torch.nn.Module):
    """`FastSSCD` behind a gate on REAL data: the bf16 configuration is accuracy-tested on a calibrated
    random-init trunk (tests/test_inference.py), which says nothing about a particular checkpoint.  On the first batch
    of frames the fp32 eager network runs as well; if any frame's descriptors agree worse than `min_cosine` the fast
    network is dropped for the rest of the run (with a warning) and the eager one answers."""

    def __init__(self, fast: torch.nn.Module, eager: torch.nn.Module, min_cosine: float = 0.999):
        super().__init__()
        self.fast, self.eager, self.min_cosine = fast, eager, float(min_cosine)
        self.checked, self.use_fast, self.first_batch_cosine = False, True, None

    def forward(self, x):
        if not self.use_fast:
            return self.eager(x)
        y = self.fast(x)
        if not self.checked:
            self.checked = True
            ref = self.eager(x)
            cos = torch.nn.functional.cosine_similarity(ref.float(), y.float(), dim=1)
            self.first_batch_cosine = float(cos.min().item()) if cos.numel() else 1.0
            if not (self.first_batch_cosine >= self.min_cosine):
                logger.warning(f"--fast: descriptors of the first batch agree with the fp32 network only to cosine "
                               f"{self.first_batch_cosine:.5f} (< {self.min_cosine}); continuing on the fp32 network")
                self.use_fast = False
                return ref
        return y


def load_model(args, device):
    if args.torchscript_path:
        model = torch.jit.load(args.torchscript_path, map_location=device)
    else:
        logger.warning("no --torchscript_path: RANDOM-INIT network of the SSCD architecture (benchmarks only)")
        model = build_sscd_model(device=device, channels_last=False)
    model = model.eval().to(device)
    if getattr(args, "fast", False):
        # (not a reference flag) the same weights through FastSSCD: bf16 NHWC trunk, GEMM 1x1, fused epilogues
        from vsc2022_amd.vsc.baseline.inference import FastSSCD, SSCDModel, sscd_from_module

        if device.type != "cuda":
            raise Exception("--fast needs --accelerator cuda")
        eager = model if isinstance(model, SSCDModel) else sscd_from_module(model)
        if eager is None:
            raise Exception("--fast: the model is not a ResNet-50 trunk + GeM + Linear (or does not reproduce on a "
                            "random batch after conversion); run without --fast")
        model = CheckedFast(FastSSCD(eager).to(device).eval(), eager, getattr(args, "fast_min_cosine", 0.999)).eval()
    return model


def worker_process(args, rank: int, world_size: int, output_filename: str):
    logger.info(f"Starting worker {rank} of {world_size}.")
    device = get_device(args, rank, world_size)
    model = load_model(args, device)
    dataset = VideoDataset(args.dataset_path, fps=args.fps, extensions=args.video_extensions.split(","),
                           distributed_rank=rank, distributed_world_size=world_size,
                           video_reader=VideoReaderType[args.video_reader.upper()], ffmpeg_path=args.ffmpeg_path)
    vfs = list(run_inference(dataset, model, device, InferenceTransforms[args.transforms], args.batch_size,
                             args.store_fp16))
    store_features(output_filename, vfs)
    logger.info(f"Wrote worker {rank} features for {len(vfs)} videos to {output_filename}")


def _spawned(rank: int, args, world_size: int, worker_files: List[str]):
    worker_process(args, rank, world_size, worker_files[rank])


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="descriptor inference on PyTorch-ROCm")
    g = p.add_argument_group("Inference")
    g.add_argument("--baseline", default="sscd", choices=[x.name.lower() for x in Baseline])
    g.add_argument("--torchscript_path", default=None)
    g.add_argument("--batch_size", type=int, default=32)
    g.add_argument("--distributed_rank", type=int, default=0)
    g.add_argument("--distributed_size", type=int, default=1)
    g.add_argument("--processes", type=int, default=1)
    g.add_argument("--transforms", choices=[x.name for x in InferenceTransforms], default="RESIZE_320_CENTER")
    g.add_argument("--accelerator", choices=[x.name.lower() for x in Accelerator], default="cpu")
    g.add_argument("--output_file", required=True)
    g.add_argument("--scratch_path", required=False)
    g.add_argument("--store_fp16", action="store_true")
    g.add_argument("--fast", action="store_true",
                   help="(extension) run a ResNet-50 SSCD model through FastSSCD: bf16 trunk, fused kernels; cuda only")
    g.add_argument("--fast_min_cosine", type=float, default=0.999,
                   help="(extension) --fast: the first batch is also run on the fp32 network; below this per-frame "
                        "cosine the run continues on the fp32 network")
    d = p.add_argument_group("Dataset")
    d.add_argument("--dataset_path", required=True)
    d.add_argument("--fps", default=1, type=float)
    d.add_argument("--video_extensions", default="mp4")
    d.add_argument("--video_reader", choices=[x.name for x in VideoReaderType], default="FFMPEG")
    d.add_argument("--ffmpeg_path", default="ffmpeg")
    return p


def main(args):
    if args.processes > 1 and args.distributed_size > 1:
        raise Exception("Set either --processes (single-machine distributed) or both --distributed_size and "
                        "--distributed_rank (arbitrary distributed)")
    with tempfile.TemporaryDirectory() as tmp_path:
        out_dir = os.path.dirname(args.output_file)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        scratch = args.scratch_path or tmp_path
        os.makedirs(scratch, exist_ok=True)
        if args.processes > 1:
            import torch.multiprocessing as mp

            worker_files = [os.path.join(scratch, f"{rank}.npz") for rank in range(args.processes)]
            logger.info(f"Spawning {args.processes} processes")
            mp.spawn(_spawned, args=(args, args.processes, worker_files), nprocs=args.processes, join=True)
            n = merge_feature_files(worker_files, args.output_file)
            logger.info(f"Features for {n} videos saved to {args.output_file}")
        else:
            worker_process(args, args.distributed_rank, args.distributed_size, args.output_file)


if __name__ == "__main__":
    logging.basicConfig(format="%(asctime)s %(levelname)-8s %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S")
    main(build_parser().parse_args())
