#!/usr/bin/env python3
"""Config 3 (frame inference) variants on one GPU, forward passes of a fixed packed batch: autocast vs a trunk held in
bf16, MIOpen's find mode (torch.backends.cudnn.benchmark), batch size.  Prints frames/s per variant and the cosine of
every variant's descriptors against the fp32 eager network on the same frames."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vsc2022_amd.vsc.baseline.inference import build_sscd_model, fold_batchnorm, preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--batches", type=int, nargs="+", default=[256, 512])
args = ap.parse_args()
dev = torch.device("cuda", 0)
model = build_sscd_model(device=dev)
fused = fold_batchnorm(model).to(memory_format=torch.channels_last)
g = torch.Generator(device=dev)
g.manual_seed(5)


def trunk_in(m, dtype):
    import copy

    h = copy.deepcopy(m)
    h.stem.to(dtype)
    h.trunk.to(dtype)
    return h.to(memory_format=torch.channels_last)


@torch.no_grad()
def timed(name, net, x, amp=None, iters=args.iters):
    def fwd():
        if amp is not None:
            with torch.autocast("cuda", dtype=amp):
                return net(x)
        return net(x)

    for _ in range(2):
        y = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        y = fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return name, x.shape[0] / dt, y.float()


for find in (False, True):
    torch.backends.cudnn.benchmark = find
    for bs in args.batches:
        u8 = torch.randint(0, 256, (bs, 3, 320, 320), generator=g, device=dev, dtype=torch.uint8)
        x = preprocess(u8)
        with torch.no_grad():
            ref = model(x[:64]).float()
        rows = [timed("folded, autocast bf16", fused, x, torch.bfloat16),
                timed("folded, trunk in bf16", trunk_in(fused, torch.bfloat16), x.to(torch.bfloat16)),
                timed("folded, trunk in fp16", trunk_in(fused, torch.float16), x.to(torch.float16)),
                timed("folded, fp32", fused, x, None, iters=2)]
        for name, fps, y in rows:
            cos = F.cosine_similarity(y[:64], ref, dim=1).min().item()
            print(f"find={int(find)} batch {bs:4d}  {name:24s} {fps:9.1f} frames/s   min cosine vs fp32 eager {cos:.5f}", flush=True)
