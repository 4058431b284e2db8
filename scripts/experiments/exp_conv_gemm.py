#!/usr/bin/env python3
"""`vsc_conv_bias_act_bf16` on the trunk's convolution shapes at batch 256: error against an fp32 convolution of the
same bf16 values, time against MIOpen + the epilogue pass (3x3) and against FastSSCD's 1x1 paths."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vsc2022_amd.vsc.baseline.inference import _bias_act, _conv_bias_act, _gemm_bias_act, _rows

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(2)
B = int(os.environ.get("BATCH", "256"))


def timed(fn, it=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


# correctness on small cases first (odd sizes, both strides, identity, 1x1)
for (b, c, h, w, n, k, s, has_res, relu) in ((2, 64, 5, 7, 64, 3, 1, False, True), (3, 128, 9, 6, 128, 3, 2, True, True), (1, 64, 1, 1, 64, 3, 1, False, False),
                                             (2, 64, 12, 12, 256, 1, 1, True, True), (5, 192, 17, 13, 64, 3, 2, False, True), (2, 256, 8, 8, 512, 1, 1, False, False)):
    x = cl((torch.randn((b, c, h, w), generator=g, device=dev)).to(torch.bfloat16))
    wt = cl((torch.randn((n, c, k, k), generator=g, device=dev) / (k * k * c) ** 0.5).to(torch.bfloat16))
    bias = torch.randn(n, generator=g, device=dev)
    ref = F.conv2d(x.float(), wt.float(), bias, s, k // 2)
    res = cl(torch.randn(ref.shape, generator=g, device=dev).to(torch.bfloat16)) if has_res else None
    if has_res:
        ref = ref + res.float()
    if relu:
        ref = ref.relu()
    got = _conv_bias_act(x, wt, bias, res, s, relu).float()
    err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
    print(f"check B={b} C={c} {h}x{w} N={n} k={k} s={s} res={has_res} relu={relu}: shape ok {tuple(got.shape) == tuple(ref.shape)}  max rel err {err:.2e}", flush=True)

tot_new = tot_old = 0.0
for C, H, s, cnt in ((64, 80, 1, 3), (128, 80, 2, 1), (128, 40, 1, 3), (256, 40, 2, 1), (256, 20, 1, 5), (512, 20, 2, 1), (512, 10, 1, 2)):
    x = cl(torch.randn((B, C, H, H), generator=g, device=dev).to(torch.bfloat16))
    wt = cl((torch.randn((C, C, 3, 3), generator=g, device=dev) / (9 * C) ** 0.5).to(torch.bfloat16))
    bias = torch.randn(C, generator=g, device=dev)

    def old():
        y = F.conv2d(x, wt, None, s, 1)
        return _bias_act(_rows(y), bias, None, True)

    t_new, t_old = timed(lambda: _conv_bias_act(x, wt, bias, None, s, True)), timed(old)
    flop = 2 * B * ((H - 1) // s + 1) ** 2 * C * C * 9
    tot_new += cnt * t_new
    tot_old += cnt * t_old
    print(f"3x3 C={C:4d} H={H:3d} stride {s}: new {t_new:6.3f} ms ({flop / t_new / 1e9:5.0f} TFLOP/s)   MIOpen + epilogue pass {t_old:6.3f} ms   (x{cnt} per forward pass)", flush=True)
print(f"3x3 convolutions of one forward pass: new {tot_new:.2f} ms, before {tot_old:.2f} ms")

tot_new = tot_old = 0.0
for H, K, N, has_res, relu, cnt, what in ((80, 256, 64, False, True, 2, "l1 conv1"), (80, 256, 128, False, True, 1, "l2.0 conv1"), (40, 512, 128, False, True, 3, "l2 conv1"),
                                          (40, 256, 512, False, False, 1, "l2 down"), (40, 512, 256, False, True, 1, "l3.0 conv1"), (20, 1024, 256, False, True, 5, "l3 conv1"),
                                          (20, 256, 1024, True, True, 6, "l3 conv3"), (20, 512, 1024, False, False, 1, "l3 down"), (20, 1024, 512, False, True, 1, "l4.0 conv1"),
                                          (10, 2048, 512, False, True, 2, "l4 conv1"), (10, 512, 2048, True, True, 3, "l4 conv3"), (10, 1024, 2048, False, False, 1, "l4 down"),
                                          (80, 64, 256, True, True, 3, "l1 conv3"), (40, 128, 512, True, True, 4, "l2 conv3")):
    x = cl(torch.randn((B, K, H, H), generator=g, device=dev).to(torch.bfloat16))
    wt = cl((torch.randn((N, K, 1, 1), generator=g, device=dev) / K ** 0.5).to(torch.bfloat16))
    bias = torch.randn(N, generator=g, device=dev)
    res = cl(torch.randn((B, N, H, H), generator=g, device=dev).to(torch.bfloat16)) if has_res else None
    a2, w2, wt2, bh = _rows(x), wt.reshape(N, K), wt.reshape(N, K).t().contiguous(), bias.to(torch.bfloat16)
    r2 = _rows(res) if has_res else None

    def old():
        if K <= 128:
            return _gemm_bias_act(a2, w2, bias, r2, relu)
        if not has_res and relu:
            return torch._addmm_activation(bh, a2, wt2)
        return _bias_act(torch.mm(a2, wt2), bias, r2, relu)

    t_new, t_old = timed(lambda: _conv_bias_act(x, wt, bias, res, 1, relu)), timed(old)
    tot_new += cnt * t_new
    tot_old += cnt * t_old
    print(f"1x1 {what:11s} K={K:4d} N={N:4d} H={H:3d}: new {t_new:6.3f} ms   FastSSCD's current path {t_old:6.3f} ms   (x{cnt})", flush=True)
print(f"1x1 convolutions of one forward pass: new {tot_new:.2f} ms, before {tot_old:.2f} ms")
