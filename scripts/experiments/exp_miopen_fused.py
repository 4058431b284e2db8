#!/usr/bin/env python3
"""Is MIOpen's fused conv + bias + ReLU (aten::miopen_convolution_relu) usable for the trunk's 3x3 convolutions?"""
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)


def timed(fn, it=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for dt in (torch.float16, torch.bfloat16):
    for C, H, s in ((64, 80, 1), (128, 40, 1), (256, 20, 1), (512, 10, 1), (128, 80, 2)):
        x = torch.randn((256, C, H, H), generator=g, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((C, C, 3, 3), generator=g, device=dev) / (9 * C) ** 0.5).to(dt).contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, generator=g, device=dev).to(dt)
        t_plain = timed(lambda: F.conv2d(x, w, None, s, 1))
        t_bias = timed(lambda: torch.relu_(F.conv2d(x, w, b, s, 1)))
        try:
            t_fused = timed(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [1, 1], [1, 1], 1), it=2)
        except Exception as e:  # noqa: BLE001
            t_fused = float("nan")
            print("fused failed:", str(e)[:100])
        flop = 2 * 256 * (H // s) ** 2 * C * C * 9
        print(f"{str(dt)[6:]:9s} C={C:4d} H={H:3d} stride {s}: conv {t_plain:7.3f} ms ({flop / t_plain / 1e9:6.0f} TFLOP/s)   conv+bias, relu_ {t_bias:7.3f} ms   "
              f"miopen_convolution_relu {t_fused:9.3f} ms", flush=True)
