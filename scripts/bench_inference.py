#!/usr/bin/env python3
"""BASELINE config 3: SSCD-shaped ResNet-50 @1fps frame inference on synthetic videos, PyTorch-ROCm,
one GPU.  Reports frames/s and videos/s for fp32 and bf16/fp16 autocast (channels-last), per-video
batches (the reference's batching) and packed batches."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vsc2022_amd.vsc.baseline.inference import (SyntheticVideos, build_sscd_model, fold_batchnorm, run_inference,
                                                run_inference_packed)

ap = argparse.ArgumentParser()
ap.add_argument("--videos", type=int, default=512)
ap.add_argument("--frames", type=int, default=25)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--packed-batch", type=int, default=128)
ap.add_argument("--fold-bn", action="store_true", help="also time the network with its BatchNorms folded into the convolutions")
ap.add_argument("--fast-only", action="store_true", help="only the two fast configurations: folded + bf16 autocast, and FastSSCD")
args = ap.parse_args()
dev = torch.device("cuda", 0)
model = build_sscd_model(device=dev)
src = SyntheticVideos(n_videos=args.videos, frames=(args.frames, args.frames), size=320)
warm = SyntheticVideos(n_videos=16, frames=(args.frames, args.frames), size=320)
if args.fast_only:
    from vsc2022_amd.vsc.baseline.inference import FastSSCD

    fused = fold_batchnorm(model).to(memory_format=torch.channels_last)
    for name, net, dt in (("bf16-autocast  packed, BN folded", fused, torch.bfloat16),
                          ("FastSSCD (bf16 trunk, GEMM 1x1, fused epilogues)", FastSSCD(model).to(dev), None)):
        for bs in (256, 512):
            for _ in run_inference_packed(net, warm, dev, bs, dt):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _, d in run_inference_packed(net, src, dev, bs, dt):
                n += d.shape[0]
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            print(f"{name}, batch<={bs:4d}: {n} frames in {dtm:.2f} s = {n / dtm:8.1f} frames/s {args.videos / dtm:7.1f} videos/s",
                  flush=True)
    sys.exit(0)
for name, dt in (("fp32", None), ("bf16-autocast", torch.bfloat16), ("fp16-autocast", torch.float16)):
    for mode, fn, bs in (("per-video", run_inference, args.batch), ("packed", run_inference_packed, args.packed_batch)):
        for _ in fn(model, warm, dev, bs, dt):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _, d in fn(model, src, dev, bs, dt):
            n += d.shape[0]
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        print(f"{name:14s} {mode:9s} batch<={bs:4d}: {n} frames in {dtm:.2f} s = {n / dtm:8.1f} frames/s "
              f"{args.videos / dtm:7.1f} videos/s")
if args.fold_bn:
    fused = fold_batchnorm(model).to(memory_format=torch.channels_last)
    for name, dt in (("bf16-autocast", torch.bfloat16), ("fp16-autocast", torch.float16)):
        for bs in (128, 256):
            for _ in run_inference_packed(fused, warm, dev, bs, dt):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _, d in run_inference_packed(fused, src, dev, bs, dt):
                n += d.shape[0]
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t0
            print(f"{name:14s} packed, BN folded, batch<={bs:4d}: {n} frames in {dtm:.2f} s = {n / dtm:8.1f} frames/s "
                  f"{args.videos / dtm:7.1f} videos/s")
