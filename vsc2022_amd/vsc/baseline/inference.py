"""Frame-descriptor inference on PyTorch-ROCm (BASELINE config 3).

Mirror of the *contract* of the reference's `vsc/baseline/inference.py` / `inference_impl.py`
(paths relative to /root/reference): frames at 1 fps -> [B, 3, 320, 320] normalised tensors
(`inference_impl.py:39-69`) -> model -> one [n_frames, 512] descriptor block per video with
`(i/fps, (i+1)/fps)` timestamps (`video_reader/ffmpeg_video_reader.py:54`), videos partitioned over
ranks by `video_idx % world_size == rank` (`inference_impl.py:105-109`), batches of <= 32 frames of
a single video (`inference.py:58,65`, `inference_impl.py:210-239`).

What differs, deliberately:
  * no ffmpeg / TorchScript file: there is neither a decoder nor SSCD weights in this environment
    (no network), so frames come from a seeded synthetic source and the model is a random-init
    network of the SSCD architecture (ResNet-50 trunk + GeM + Linear(2048 -> 512), the trailing
    L2-norm removed as `adapt_sscd_model.py:54-77` does);
  * this is stock PyTorch-ROCm (MIOpen convolutions): the north-star keeps frame inference off
    the hand-written-kernel path;
  * descriptors can be handed to the matching engine as device tensors (no `.npz` round trip,
    no `.cpu()` per batch as in `inference_impl.py:228-229`).
"""
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import os
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from vsc2022_amd.vsc.index import VideoFeature

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.down = None
        if stride != 1 or inplanes != planes * 4:
            self.down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                      nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.down is None else self.down(x)
        x = F.relu(self.bn1(self.conv1(x)), inplace=True)
        x = F.relu(self.bn2(self.conv2(x)), inplace=True)
        x = self.bn3(self.conv3(x))
        return F.relu(x + idt, inplace=True)


class SSCDModel(nn.Module):
    """ResNet-50 trunk -> GeM(p=3) -> Linear(2048 -> dims); no final L2 normalisation."""

    def __init__(self, dims: int = 512, gem_p: float = 3.0):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
                                  nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1))
        layers, inplanes = [], 64
        for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            for b in range(blocks):
                layers.append(Bottleneck(inplanes, planes, stride if b == 0 else 1))
                inplanes = planes * 4
        self.trunk = nn.Sequential(*layers)
        self.gem_p = gem_p
        self.embed = nn.Linear(2048, dims)

    def forward(self, x):
        x = self.trunk(self.stem(x))
        x = x.float().clamp(min=1e-6).pow(self.gem_p).mean(dim=(2, 3)).pow(1.0 / self.gem_p)
        return self.embed(x)


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """conv followed by an eval-mode BatchNorm as ONE conv with bias: w' = w * g / sqrt(var + eps),
    b' = beta - mean * g / sqrt(var + eps)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
    fused.weight.data = (conv.weight * scale.view(-1, 1, 1, 1)).detach().clone()
    fused.bias.data = (bn.bias - bn.running_mean * scale).detach().clone()
    return fused.to(conv.weight.device, conv.weight.dtype)


class _Identity(nn.Module):
    def forward(self, x):
        return x


def fold_batchnorm(model: "SSCDModel") -> "SSCDModel":
    """Inference-only rewrite: every BatchNorm of the trunk folded into the convolution before it (the BN kernels are
    pure HBM traffic: a third of the elementwise passes of a ResNet-50 forward).  Same function up to fp rounding."""
    import copy

    m = copy.deepcopy(model).eval()
    m.stem[0], m.stem[1] = _fold(m.stem[0], m.stem[1]), _Identity()
    for blk in m.trunk:
        blk.conv1, blk.bn1 = _fold(blk.conv1, blk.bn1), _Identity()
        blk.conv2, blk.bn2 = _fold(blk.conv2, blk.bn2), _Identity()
        blk.conv3, blk.bn3 = _fold(blk.conv3, blk.bn3), _Identity()
        if blk.down is not None:
            blk.down = nn.Sequential(_fold(blk.down[0], blk.down[1]))
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def _bias_act(y2d: torch.Tensor, bias: torch.Tensor, res2d: Optional[torch.Tensor], relu: bool) -> torch.Tensor:
    """In place y = act(y + bias (+ res)) on [rows, channels] bf16 device matrices: `vsc_bias_act_bf16`
    (csrc/eltwise.hip), on torch's current stream."""
    from vsc2022_amd import _lib

    assert y2d.is_cuda and y2d.dtype == torch.bfloat16 and y2d.is_contiguous() and bias.dtype == torch.float32
    assert res2d is None or (res2d.dtype == torch.bfloat16 and res2d.is_contiguous() and res2d.shape == y2d.shape)
    _lib.check(_lib.lib().vsc_bias_act_bf16(y2d.data_ptr(), 0 if res2d is None else res2d.data_ptr(), bias.data_ptr(),
                                            y2d.shape[0], y2d.shape[1], 1 if relu else 0,
                                            torch.cuda.current_stream(y2d.device).cuda_stream))
    return y2d


def _gemm_bias_act(a2d: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, res2d: Optional[torch.Tensor], relu: bool):
    """act(a2d @ w.T + bias (+ res)) as one kernel: `vsc_gemm_bias_act_bf16` (csrc/gemm_epi.hip); a2d [M, K], w [N, K],
    res2d [M, N] bf16 on the device, bias fp32."""
    from vsc2022_amd import _lib

    assert a2d.is_cuda and a2d.dtype == w.dtype == torch.bfloat16 and a2d.is_contiguous() and w.is_contiguous()
    assert bias.dtype == torch.float32 and a2d.shape[1] == w.shape[1] and bias.shape[0] == w.shape[0]
    assert res2d is None or (res2d.dtype == torch.bfloat16 and res2d.is_contiguous() and tuple(res2d.shape) == (a2d.shape[0], w.shape[0]))
    out = torch.empty((a2d.shape[0], w.shape[0]), dtype=torch.bfloat16, device=a2d.device)
    _lib.check(_lib.lib().vsc_gemm_bias_act_bf16(a2d.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                                 0 if res2d is None else res2d.data_ptr(), out.data_ptr(), a2d.shape[0],
                                                 w.shape[0], a2d.shape[1], 1 if relu else 0,
                                                 torch.cuda.current_stream(a2d.device).cuda_stream))
    return out


def _conv_bias_act(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, res: Optional[torch.Tensor], stride: int, relu: bool):
    """act(conv(x, w) + bias (+ res)) for a 3x3 / padding 1 (or 1x1) convolution as one kernel: `vsc_conv_bias_act_bf16`
    (csrc/conv_gemm.hip).  x [B, C, H, W] and w [N, C, kh, kw] bf16 in channels-last memory, res / result [B, N, Ho, Wo]
    likewise; bias fp32."""
    from vsc2022_amd import _lib

    b, c, h, wd = x.shape
    n, _, kh, kw = w.shape
    assert x.is_cuda and x.dtype == w.dtype == torch.bfloat16 and bias.dtype == torch.float32 and (kh, kw) in ((1, 1), (3, 3))
    assert x.permute(0, 2, 3, 1).is_contiguous() and w.permute(0, 2, 3, 1).is_contiguous()
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    out = torch.empty((b, ho, wo, n), dtype=torch.bfloat16, device=x.device)
    if res is not None:
        assert res.dtype == torch.bfloat16 and tuple(res.shape) == (b, n, ho, wo) and res.permute(0, 2, 3, 1).is_contiguous()
    _lib.check(_lib.lib().vsc_conv_bias_act_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr(), 0 if res is None else res.data_ptr(),
                                                 out.data_ptr(), b, h, wd, c, n, kh * kw, stride, 1 if relu else 0,
                                                 torch.cuda.current_stream(x.device).cuda_stream))
    return out.permute(0, 3, 1, 2)


def _conv1x1_rows(a2d: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, res2d: Optional[torch.Tensor], relu: bool):
    """The same kernel for a 1x1 convolution on the [M, K] view (M "images" of one pixel): a2d [M, K], w [N, K], res2d [M, N]."""
    from vsc2022_amd import _lib

    assert a2d.is_cuda and a2d.dtype == w.dtype == torch.bfloat16 and a2d.is_contiguous() and w.is_contiguous()
    assert bias.dtype == torch.float32 and a2d.shape[1] == w.shape[1]
    assert res2d is None or (res2d.dtype == torch.bfloat16 and res2d.is_contiguous() and tuple(res2d.shape) == (a2d.shape[0], w.shape[0]))
    out = torch.empty((a2d.shape[0], w.shape[0]), dtype=torch.bfloat16, device=a2d.device)
    _lib.check(_lib.lib().vsc_conv_bias_act_bf16(a2d.data_ptr(), w.data_ptr(), bias.data_ptr(), 0 if res2d is None else res2d.data_ptr(),
                                                 out.data_ptr(), a2d.shape[0], 1, 1, a2d.shape[1], w.shape[0], 1, 1, 1 if relu else 0,
                                                 torch.cuda.current_stream(a2d.device).cuda_stream))
    return out


def _pool_bias_relu(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """maxpool3x3/2(relu(x + bias)) of an NHWC bf16 tensor in one pass: `vsc_pool3x3s2_bias_relu_bf16` (csrc/eltwise.hip)."""
    from vsc2022_amd import _lib

    n, c, h, w = x.shape
    xl = x.permute(0, 2, 3, 1)
    assert x.is_cuda and x.dtype == torch.bfloat16 and xl.is_contiguous() and bias.dtype == torch.float32
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty((n, ho, wo, c), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().vsc_pool3x3s2_bias_relu_bf16(xl.data_ptr(), bias.data_ptr(), out.data_ptr(), n, h, w, c,
                                                       torch.cuda.current_stream(x.device).cuda_stream))
    return out.permute(0, 3, 1, 2)


def _rows(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] in channels-last memory -> the [N*H*W, C] matrix over the same bytes."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c)


# conv1's bias + ReLU inside hipBLASLt's epilogue (`torch._addmm_activation`, bf16 bias): +4 % on the forward pass.
# VSC_FAST_GEMM_EPILOGUE=0: a plain GEMM followed by `vsc_bias_act_bf16` like the other convolutions (fp32 bias).
# Measured and dropped: `aten::miopen_convolution_relu` for conv2 (falls onto a path 200x slower here); `addmm` with the
# identity as its C matrix (torch copies C into the output first: a whole extra pass).
_GEMM_EPILOGUE = os.environ.get("VSC_FAST_GEMM_EPILOGUE", "1") != "0"
_FUSED_POOL = os.environ.get("VSC_FAST_FUSED_POOL", "1") != "0"  # the stem's bias + ReLU + max-pool as one kernel
# the 3x3 convolutions and the 1x1 shapes of _CONV_KERNEL_1X1 through `vsc_conv_bias_act_bf16` (0: MIOpen / hipBLASLt + passes)
_FUSED_CONV = os.environ.get("VSC_FAST_FUSED_CONV", "1") != "0"
# largest Cin for which a 1x1 convolution runs as `vsc_gemm_bias_act_bf16` (0: never); see _Conv1x1
_FUSED_GEMM_MAX_K = int(os.environ.get("VSC_FAST_FUSED_GEMM_MAX_K", "128"))  # (layer2's (128, 512) goes to the conv kernel)


# 1x1 convolutions (Cin, Cout) that run faster through the implicit-GEMM kernel (`vsc_conv_bias_act_bf16`, taps = 1) than
# through hipBLASLt + an epilogue pass, measured at batch 256 on 320 x 320 frames (profiles/r03_config3_inference.md: the
# kernel keeps its epilogue inside but reaches 0.55-0.64 PFLOP/s, so the deep, arithmetic-heavy shapes stay with hipBLASLt)
_CONV_KERNEL_1X1 = {(256, 64), (512, 128), (128, 512), (256, 512), (256, 1024), (512, 1024), (512, 2048)}


class _Conv1x1(nn.Module):
    """A folded 1x1 convolution on the [N*H*W, Cin] view, with what follows it (bias, identity, ReLU).  Three routes:
    `gemm`  Cin = 64 (layer1: ~1 GB of activations per call, hardly any arithmetic): `vsc_gemm_bias_act_bf16`
            (csrc/gemm_epi.hip), operands straight from global memory, epilogue inside;
    `conv`  the shapes of _CONV_KERNEL_1X1: `vsc_conv_bias_act_bf16` (csrc/conv_gemm.hip) with one tap, epilogue inside;
    `blas`  the rest: hipBLASLt through torch (a plain library GEMM) followed by one pass of `vsc_bias_act_bf16`, or, when
            there is no identity, with bias + ReLU in hipBLASLt's own epilogue (`torch._addmm_activation`)."""

    def __init__(self, conv: nn.Conv2d):
        super().__init__()
        bf = torch.bfloat16
        k, n = conv.in_channels, conv.out_channels
        w = conv.weight.detach().reshape(n, k)
        ok = k % 64 == 0 and n % 64 == 0
        if ok and _FUSED_CONV and (k, n) in _CONV_KERNEL_1X1:
            self.route = "conv"
        elif ok and k <= _FUSED_GEMM_MAX_K:
            self.route = "gemm"
        else:
            self.route = "blas"
        self.w = nn.Parameter((w.t() if self.route == "blas" else w).contiguous().to(bf), requires_grad=False)  # [Cin, Cout] / [Cout, Cin]
        self.bias = nn.Parameter(conv.bias.detach().float().clone(), requires_grad=False)
        self.bias_h = nn.Parameter(self.bias.detach().to(bf), requires_grad=False)

    def forward(self, x2d, res2d, relu: bool):
        if self.route == "conv":
            return _conv1x1_rows(x2d, self.w, self.bias, res2d, relu)
        if self.route == "gemm":
            return _gemm_bias_act(x2d, self.w, self.bias, res2d, relu)
        if res2d is None and relu and _GEMM_EPILOGUE:
            return torch._addmm_activation(self.bias_h, x2d, self.w)
        return _bias_act(torch.mm(x2d, self.w), self.bias, res2d, relu)


class FastBottleneck(nn.Module):
    """One bottleneck of the folded trunk, bf16 activations in NHWC memory, for inference on the GPU:
    the 1x1 convolutions are GEMMs over the [N*H*W, C] view (`_Conv1x1`; MIOpen's 1x1 kernels run them 1.3x slower),
    and what follows a convolution -- bias, identity, ReLU -- is the epilogue of that GEMM or ONE in-place pass of
    `vsc_bias_act_bf16` (csrc/eltwise.hip) instead of stock PyTorch's separate bias / add / relu passes: at batch 256
    the trunk is bound by the HBM traffic of its activations, not by its matrix products (profiles/r03_config3_inference.md)."""

    def __init__(self, blk: "Bottleneck"):
        super().__init__()
        self.c1, self.c3 = _Conv1x1(blk.conv1), _Conv1x1(blk.conv3)
        c2 = blk.conv2
        self.stride = c2.stride[0]
        self.w2 = nn.Parameter(c2.weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last),
                               requires_grad=False)
        self.b2 = nn.Parameter(c2.bias.detach().float().clone(), requires_grad=False)
        self.has_down = blk.down is not None
        if self.has_down:
            self.down_stride = blk.down[0].stride[0]
            self.cd = _Conv1x1(blk.down[0])

    def forward(self, x):
        n, _, h, w = x.shape
        x2 = _rows(x)
        if self.has_down:
            xs = x if self.down_stride == 1 else x[:, :, :: self.down_stride, :: self.down_stride].contiguous(
                memory_format=torch.channels_last)
            idt = self.cd(_rows(xs), None, False)
        else:
            idt = x2
        y = self.c1(x2, None, True)                                                       # conv1 + bias + relu
        y = y.view(n, h, w, -1).permute(0, 3, 1, 2)
        if _FUSED_CONV and y.shape[1] % 64 == 0 and self.w2.shape[0] % 64 == 0:
            y = _conv_bias_act(y, self.w2, self.b2, None, self.stride, True)              # conv2 (3x3) + bias + relu, one kernel
            n2, _, h2, w2 = y.shape
            y = _rows(y)
        else:
            y = F.conv2d(y, self.w2, None, self.stride, 1)                                # conv2 (3x3, MIOpen)
            n2, _, h2, w2 = y.shape
            y = _bias_act(_rows(y), self.b2, None, True)                                  # + bias + relu
        out = self.c3(y, idt, True)                                                       # conv3 + bias + identity + relu
        return out.view(n2, h2, w2, -1).permute(0, 3, 1, 2)


class FastSSCD(nn.Module):
    """`SSCDModel` prepared for inference on the GPU: BatchNorms folded, stem and trunk in bf16 (NHWC), blocks as
    `FastBottleneck`; GeM and the embedding stay in fp32.  Takes the fp32 frames `preprocess` returns.  The accuracy
    gate against the fp32 eager network is tests/test_inference.py::test_fast_inference_configuration_against_fp32_eager."""

    def __init__(self, model: "SSCDModel"):
        super().__init__()
        m = fold_batchnorm(model)
        self.stem_conv = m.stem[0].to(torch.bfloat16).to(memory_format=torch.channels_last)
        self.stem_bias = nn.Parameter(self.stem_conv.bias.detach().float().clone(), requires_grad=False)
        self.stem_conv.bias = None
        self.pool = m.stem[3]
        self.blocks = nn.ModuleList(FastBottleneck(b) for b in m.trunk)
        self.gem_p = m.gem_p
        self.embed = m.embed
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        # the hand-written kernels launch on torch's current stream OF THE TENSOR'S DEVICE and never switch devices
        # themselves: make that device current for the whole pass (a caller may sit on another one)
        with torch.cuda.device(x.device):
            return self._forward(x)

    def _forward(self, x):
        x = self.stem_conv(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
        if _FUSED_POOL and x.permute(0, 2, 3, 1).is_contiguous():
            x = _pool_bias_relu(x, self.stem_bias)                                        # bias + relu + max-pool, one pass
        else:
            n, _, h, w = x.shape
            x = _bias_act(_rows(x), self.stem_bias, None, True).view(n, h, w, -1).permute(0, 3, 1, 2)
            x = self.pool(x)
        for b in self.blocks:
            x = b(x)
        x = x.float().clamp(min=1e-6).pow(self.gem_p).mean(dim=(2, 3)).pow(1.0 / self.gem_p)
        return self.embed(x)


def sscd_from_module(module, check_tol: float = 1e-3) -> Optional["SSCDModel"]:
    """An `SSCDModel` holding the weights of `module` -- a TorchScript (or eager) ResNet-50 trunk + GeM + Linear such as
    the torchvision SSCD models after `adapt_sscd_model.py:54-77` removed the L2 norm -- or None when it is not that
    architecture.  Parameter NAMES differ between exports (backbone.layer1.0.downsample.0.weight, ...); their ORDER and
    shapes are those of a ResNet-50 (conv1, bn1, per bottleneck conv1 bn1 conv2 bn2 conv3 bn3 [downsample conv bn],
    projection), so the tensors are matched by position and shape and the result is verified: both networks must agree
    on a random batch (squared distance of the descriptors <= check_tol, the reference's own criterion in
    adapt_sscd_model.py:45-51) or None is returned.  This is what lets `FastSSCD` run real SSCD weights."""
    try:
        src = [(k, v) for k, v in module.state_dict().items() if not k.endswith("num_batches_tracked")]
    except Exception:
        return None
    if not src or src[-1][1].dim() != 1 or src[-2][1].dim() != 2:
        return None
    dims = int(src[-2][1].shape[0])
    model = SSCDModel(dims).eval()
    dst = [(k, v) for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")]
    if len(src) != len(dst) or any(a[1].shape != b[1].shape for a, b in zip(src, dst)):
        return None
    with torch.no_grad():
        for (_, a), (_, b) in zip(src, dst):
            b.copy_(a.detach().to(device=b.device, dtype=b.dtype))
        dev = src[0][1].device
        model = model.to(dev)
        g = torch.Generator().manual_seed(0)
        x = torch.randn((2, 3, 64, 64), generator=g).to(dev)
        try:
            d = (module(x).float() - model(x).float()).pow(2).sum(dim=1)
        except Exception:
            return None
    if not bool(torch.isfinite(d).all()) or float(d.max()) > check_tol:
        return None
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def build_sscd_model(dims: int = 512, seed: int = 0, device="cpu", channels_last: bool = True) -> SSCDModel:
    torch.manual_seed(seed)
    model = SSCDModel(dims).eval().to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    for p in model.parameters():
        p.requires_grad_(False)
    return model


@dataclass
class SyntheticVideos:
    """Seeded stand-in for the ffmpeg reader: `n_videos` videos of `frames` (min, max) uint8 frames."""

    n_videos: int
    frames: Tuple[int, int] = (25, 25)
    size: int = 320
    fps: float = 1.0
    seed: int = 2
    prefix: str = "Q"

    def lengths(self) -> np.ndarray:
        rng = np.random.default_rng(self.seed)
        lo, hi = self.frames
        return rng.integers(lo, hi + 1, self.n_videos) if hi > lo else np.full(self.n_videos, lo)

    def video(self, idx: int, n_frames: int, device) -> torch.Tensor:
        g = torch.Generator(device=device)
        g.manual_seed(self.seed * 1000003 + idx)
        return torch.randint(0, 256, (n_frames, 3, self.size, self.size), generator=g, device=device,
                             dtype=torch.uint8)

    def timestamps(self, n_frames: int) -> np.ndarray:
        i = np.arange(n_frames, dtype=np.float32)
        return np.stack([i / self.fps, (i + 1) / self.fps], axis=1).astype(np.float32)


def preprocess(frames_u8: torch.Tensor, channels_last: bool = True) -> torch.Tensor:
    """ToTensor + Normalize of `inference_impl.py:50-58` on the device (frames already 320x320)."""
    x = frames_u8.float().div_(255.0)
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1)
    x = (x - mean) / std
    return x.contiguous(memory_format=torch.channels_last) if channels_last else x


@torch.no_grad()
def run_inference(model: nn.Module, source: SyntheticVideos, device, batch_size: int = 32,
                  autocast_dtype: Optional[torch.dtype] = None, rank: int = 0, world_size: int = 1,
                  channels_last: bool = True) -> Iterator[Tuple[int, torch.Tensor]]:
    """Yields (video index, [n_frames, dims] fp32 descriptors on `device`) for this rank's videos."""
    assert rank < world_size
    lengths = source.lengths()
    use_amp = autocast_dtype is not None and torch.device(device).type == "cuda"
    for idx in range(source.n_videos):
        if idx % world_size != rank:
            continue
        frames = source.video(idx, int(lengths[idx]), device)
        outs = []
        for b0 in range(0, frames.shape[0], batch_size):  # batches never mix videos (inference.py:58)
            x = preprocess(frames[b0 : b0 + batch_size], channels_last)
            if use_amp:
                with torch.autocast("cuda", dtype=autocast_dtype):
                    y = model(x)
            else:
                y = model(x)
            outs.append(y.float())
        yield idx, torch.cat(outs, dim=0)


@torch.no_grad()
def run_inference_packed(model: nn.Module, source: SyntheticVideos, device, batch_size: int = 128,
                         autocast_dtype: Optional[torch.dtype] = None, rank: int = 0, world_size: int = 1,
                         channels_last: bool = True) -> Iterator[Tuple[int, torch.Tensor]]:
    """Same results as run_inference, but frames of consecutive videos share a batch.

    The reference never mixes videos in a batch (an artefact of its per-video DataLoader); in eval
    mode the network treats every frame independently, so packing 25-frame videos into batches of
    128 only changes MIOpen's efficiency (measured on MI355X, bf16: 5.4 k -> 9.0 k frames/s).
    """
    assert rank < world_size
    lengths = source.lengths()
    use_amp = autocast_dtype is not None and torch.device(device).type == "cuda"
    pending: List[Tuple[int, int]] = []   # (video idx, n_frames) in batch order
    frames: List[torch.Tensor] = []
    done: List[torch.Tensor] = []         # descriptor blocks not yet assigned to a video

    def flush(n_take):
        batch = torch.cat(frames, dim=0)
        x = preprocess(batch[:n_take], channels_last)
        if use_amp:
            with torch.autocast("cuda", dtype=autocast_dtype):
                y = model(x)
        else:
            y = model(x)
        rest = batch[n_take:]
        frames.clear()
        if rest.shape[0]:
            frames.append(rest)
        done.append(y.float())

    def drain():
        have = sum(d.shape[0] for d in done)
        while pending and pending[0][1] <= have:
            idx, n = pending.pop(0)
            buf = torch.cat(done, dim=0)
            done.clear()
            if buf.shape[0] > n:
                done.append(buf[n:])
            have -= n
            yield idx, buf[:n]

    for idx in range(source.n_videos):
        if idx % world_size != rank:
            continue
        n = int(lengths[idx])
        frames.append(source.video(idx, n, device))
        pending.append((idx, n))
        while sum(f.shape[0] for f in frames) >= batch_size:
            flush(batch_size)
            yield from drain()
    if frames and sum(f.shape[0] for f in frames):
        flush(sum(f.shape[0] for f in frames))
    yield from drain()


def to_video_features(results, source: SyntheticVideos) -> List[VideoFeature]:
    """Host-side VideoFeature list (what `store_features` / the `.npz` route expects)."""
    out = []
    for idx, desc in results:
        out.append(VideoFeature(video_id=f"{source.prefix}{idx:06d}", timestamps=source.timestamps(desc.shape[0]),
                                feature=desc.cpu().numpy()))
    return out


def to_flat(results) -> Tuple[torch.Tensor, np.ndarray, List[int]]:
    """Device-resident hand-off into the matching engine: (features [rows, dims], row offsets, video idx)."""
    idxs, blocks = [], []
    for idx, desc in results:
        idxs.append(idx)
        blocks.append(desc)
    lens = np.array([b.shape[0] for b in blocks], dtype=np.int64)
    off = np.zeros(len(blocks) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    feats = torch.cat(blocks, dim=0) if blocks else torch.zeros((0, 0))
    return feats, off, idxs
