#!/usr/bin/env python3
"""Config 3 experiment: the 1x1 convolutions of the folded SSCD trunk as plain GEMMs (F.linear on the NHWC view,
hipBLASLt) instead of MIOpen convolutions.  Prints frames/s and the cosine against the fp32 eager network."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from vsc2022_amd.vsc.baseline.inference import build_sscd_model, fold_batchnorm, preprocess


class Gemm1x1(nn.Module):
    def __init__(self, conv: nn.Conv2d, relu: bool = False):
        super().__init__()
        assert conv.kernel_size == (1, 1) and conv.padding == (0, 0)
        self.stride = conv.stride[0]
        self.weight = nn.Parameter(conv.weight.detach().reshape(conv.out_channels, conv.in_channels).clone(), requires_grad=False)
        self.bias = nn.Parameter(conv.bias.detach().clone(), requires_grad=False)

    def forward(self, x):
        if self.stride != 1:
            x = x[:, :, :: self.stride, :: self.stride]
        n, c, h, w = x.shape
        y = F.linear(x.permute(0, 2, 3, 1).reshape(n * h * w, c), self.weight, self.bias)
        return y.view(n, h, w, -1).permute(0, 3, 1, 2)


def with_gemms(m):
    m = copy.deepcopy(m)
    for blk in m.trunk:
        blk.conv1 = Gemm1x1(blk.conv1)
        blk.conv3 = Gemm1x1(blk.conv3)
        if blk.down is not None:
            blk.down = nn.Sequential(Gemm1x1(blk.down[0]))
    return m


dev = torch.device("cuda", 0)
model = build_sscd_model(device=dev)
fused = fold_batchnorm(model).to(memory_format=torch.channels_last)
g = torch.Generator(device=dev)
g.manual_seed(5)
u8 = torch.randint(0, 256, (256, 3, 320, 320), generator=g, device=dev, dtype=torch.uint8)
x = preprocess(u8)


@torch.no_grad()
def timed(name, net, xin, amp=None, iters=8):
    def fwd():
        if amp is not None:
            with torch.autocast("cuda", dtype=amp):
                return net(xin)
        return net(xin)

    for _ in range(2):
        y = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        y = fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    cos = F.cosine_similarity(y[:64].float(), ref, dim=1).min().item()
    print(f"{name:44s} {xin.shape[0] / dt:9.1f} frames/s   min cosine vs fp32 eager {cos:.5f}", flush=True)


with torch.no_grad():
    ref = model(x[:64]).float()
gem = with_gemms(fused)
if os.environ.get("FAST"):
    from vsc2022_amd.vsc.baseline.inference import FastSSCD

    fast = FastSSCD(model).to(dev)
    timed("folded, autocast bf16 (MIOpen 1x1)", fused, x, torch.bfloat16)
    timed("FastSSCD (GEMM 1x1 + fused epilogues)", fast, x)
    from torch.profiler import ProfilerActivity, profile

    with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            fast(x)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
    sys.exit(0)
if os.environ.get("ONLY_GEMM"):
    h = copy.deepcopy(gem)
    h.stem.to(torch.bfloat16)
    h.trunk.to(torch.bfloat16)
    xb = x.to(torch.bfloat16)
    timed("folded, trunk in bf16, 1x1 as GEMM", h, xb, iters=4)
    from torch.profiler import ProfilerActivity, profile

    with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            h(xb)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
    sys.exit(0)
timed("folded, autocast bf16 (MIOpen 1x1)", fused, x, torch.bfloat16)
timed("folded, autocast bf16, 1x1 as GEMM", gem, x, torch.bfloat16)
h = copy.deepcopy(gem)
h.stem.to(torch.bfloat16)
h.trunk.to(torch.bfloat16)
timed("folded, trunk in bf16, 1x1 as GEMM", h, x.to(torch.bfloat16))
h2 = copy.deepcopy(fused)
h2.stem.to(torch.bfloat16)
h2.trunk.to(torch.bfloat16)
timed("folded, trunk in bf16 (MIOpen 1x1)", h2, x.to(torch.bfloat16))
