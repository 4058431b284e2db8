#!/usr/bin/env python3
"""`vsc_gemm_bias_act_bf16` on the 1x1-convolution shapes of the SSCD trunk at batch 256: error against fp64 and time
against what FastSSCD did before (torch.mm + the vsc_bias_act_bf16 pass; `_addmm_activation` where there is no identity)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vsc2022_amd.vsc.baseline.inference import _bias_act, _gemm_bias_act

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
B = int(os.environ.get("BATCH", "256"))
shapes = [  # (H, K, N, identity, relu, what)
    (80, 64, 64, False, True, "l1.0 conv1"), (80, 256, 64, False, True, "l1 conv1"), (80, 64, 256, True, True, "l1 conv3"),
    (80, 64, 256, False, False, "l1 down"), (80, 256, 128, False, True, "l2.0 conv1"), (40, 512, 128, False, True, "l2 conv1"),
    (40, 128, 512, True, True, "l2 conv3"), (40, 256, 512, False, False, "l2 down"), (40, 512, 256, False, True, "l3.0 conv1"),
    (20, 1024, 256, False, True, "l3 conv1"), (20, 256, 1024, True, True, "l3 conv3"), (20, 512, 1024, False, False, "l3 down"),
    (20, 1024, 512, False, True, "l4.0 conv1"), (10, 2048, 512, False, True, "l4 conv1"), (10, 512, 2048, True, True, "l4 conv3"),
    (10, 1024, 2048, False, False, "l4 down"),
]


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


tot_new = tot_old = 0.0
for H, K, N, has_res, relu, what in shapes:
    M = B * H * H
    a = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * (1.0 / K ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    res = (torch.randn((M, N), generator=g, device=dev)).to(torch.bfloat16) if has_res else None
    wt = w.t().contiguous()
    bh = bias.to(torch.bfloat16)
    out = _gemm_bias_act(a, w, bias, res, relu)
    rows = torch.randint(0, M, (2048,), generator=g, device=dev)
    ref = a[rows].double() @ w.double().t() + bias.double()
    if has_res:
        ref = ref + res[rows].double()
    if relu:
        ref = ref.relu()
    err = ((out[rows].double() - ref).abs() / (ref.abs() + 1.0)).max().item()
    last = ((out[-64:].double() - (lambda r: r.relu() if relu else r)(a[-64:].double() @ w.double().t() + bias.double() + (res[-64:].double() if has_res else 0))).abs().max().item())

    def old():
        if not has_res and relu:
            return torch._addmm_activation(bh, a, wt)
        return _bias_act(torch.mm(a, wt), bias, res, relu)

    t_new, t_old = timed(lambda: _gemm_bias_act(a, w, bias, res, relu)), timed(old)
    gb = (M * K + M * N * (2 if has_res else 1)) * 2 / 1e9
    tot_new += t_new
    tot_old += t_old
    print(f"{what:11s} M={M:8d} K={K:4d} N={N:4d}  new {t_new:6.3f} ms ({gb / t_new:5.2f} TB/s)  before {t_old:6.3f} ms   "
          f"max rel err {err:.2e} (last rows abs {last:.2e})", flush=True)
print(f"sum over the distinct shapes: new {tot_new:.2f} ms, before {tot_old:.2f} ms")
