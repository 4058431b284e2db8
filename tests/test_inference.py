"""Frame inference: model contract on CPU, device hand-off on GPU."""
import os

import numpy as np
import pytest
import torch


def test_sscd_model_contract_cpu():
    from vsc2022_amd.vsc.baseline.inference import SSCDModel, SyntheticVideos, build_sscd_model, run_inference, \
        to_video_features

    model = build_sscd_model(dims=512, seed=0, device="cpu", channels_last=False)
    n_params = sum(p.numel() for p in model.parameters())
    assert 24_000_000 < n_params < 26_000_000  # ResNet-50 trunk (23.5M) + Linear(2048 -> 512)
    src = SyntheticVideos(n_videos=5, frames=(2, 4), size=64, seed=3)
    mine = list(run_inference(model, src, "cpu", batch_size=3, rank=1, world_size=2, channels_last=False))
    assert [i for i, _ in mine] == [1, 3]                       # video_idx % world == rank
    lens = src.lengths()
    for idx, desc in mine:
        assert desc.shape == (lens[idx], 512) and desc.dtype == torch.float32 and torch.isfinite(desc).all()
    vfs = to_video_features(mine, src)
    assert vfs[0].video_id == "Q000001" and vfs[0].timestamps.shape == (lens[1], 2)
    assert np.array_equal(vfs[0].timestamps[:, 1] - vfs[0].timestamps[:, 0], np.ones(lens[1], np.float32))
    # batches never mix videos and batching does not change the result
    again = dict(run_inference(model, src, "cpu", batch_size=32, rank=1, world_size=2, channels_last=False))
    assert torch.allclose(again[1], mine[0][1], atol=1e-5)


@pytest.mark.gpu
def test_inference_feeds_the_engine_on_device(gpu):
    from vsc2022_amd.engine import DeviceMatcher
    from vsc2022_amd.vsc.baseline.inference import SyntheticVideos, build_sscd_model, run_inference, to_flat

    dev = torch.device("cuda", 0)
    model = build_sscd_model(device=dev)
    refs = SyntheticVideos(n_videos=6, frames=(4, 6), size=96, seed=5, prefix="R")
    qs = SyntheticVideos(n_videos=3, frames=(4, 6), size=96, seed=5, prefix="Q")   # same seed: videos 0..2 are copies
    rf, roff, _ = to_flat(run_inference(model, refs, dev, autocast_dtype=torch.bfloat16))
    qf, qoff, _ = to_flat(run_inference(model, qs, dev, autocast_dtype=torch.bfloat16))
    assert rf.is_cuda and rf.shape[1] == 512
    rf = rf / rf.norm(dim=1, keepdim=True)
    qf = qf / qf.norm(dim=1, keepdim=True)
    m = DeviceMatcher(rf, roff, 0)
    m.set_queries(qf, qoff)
    hi, hj, hs, _ = m.search(50)
    pq, pr, ps, _ = m.pair_max(hi, hj, hs)
    best = {}
    for a, b, s in zip(pq.cpu().tolist(), pr.cpu().tolist(), ps.cpu().tolist()):
        best.setdefault(a, (b, s))
    assert all(best[v][0] == v for v in range(3))   # every query video retrieves its own copy first


def test_packed_inference_equals_per_video_cpu():
    from vsc2022_amd.vsc.baseline.inference import SyntheticVideos, build_sscd_model, run_inference, run_inference_packed

    model = build_sscd_model(dims=32, seed=1, device="cpu", channels_last=False)
    src = SyntheticVideos(n_videos=7, frames=(1, 5), size=32, seed=4)
    a = dict(run_inference(model, src, "cpu", batch_size=4, channels_last=False))
    b = list(run_inference_packed(model, src, "cpu", batch_size=6, channels_last=False))
    assert [i for i, _ in b] == sorted(a)
    for idx, desc in b:
        assert desc.shape == a[idx].shape and torch.allclose(desc, a[idx], atol=1e-5)
    c = dict(run_inference_packed(model, src, "cpu", batch_size=6, rank=1, world_size=3, channels_last=False))
    assert sorted(c) == [1, 4] and torch.allclose(c[4], a[4], atol=1e-5)


# ---------------------------------------------------------------- transforms / CLI of inference_cli.py
@pytest.mark.parametrize("shape", [(360, 640), (480, 270), (320, 320)])
def test_device_transforms_match_pil(shape):
    """The three InferenceTransforms (inference_impl.py:39-69) as tensor ops == what torchvision does to the
    decoded PIL frame: PIL bilinear (antialiased) resize of the short edge / to a square, centre crop, /255,
    Normalize.  Resized levels may differ by one rounding step of the 0..255 value."""
    from PIL import Image

    from vsc2022_amd.vsc.baseline.inference import IMAGENET_MEAN, IMAGENET_STD
    from vsc2022_amd.vsc.baseline.inference_cli import InferenceTransforms, device_transform

    rng = np.random.default_rng(shape[0])
    h, w = shape
    # smooth image + some noise (pure noise makes every resampling-kernel difference visible)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0) for c in range(3)], axis=2)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    pil = Image.fromarray(img)
    x = torch.from_numpy(img).permute(2, 0, 1)
    mean, std = np.array(IMAGENET_MEAN, np.float32), np.array(IMAGENET_STD, np.float32)

    def ref(im):
        a = np.asarray(im, dtype=np.float32) / 255.0
        return ((a - mean) / std).transpose(2, 0, 1)

    def short_edge(size):
        return (size, int(size * w / h)) if h <= w else (int(size * h / w), size)

    th, tw = short_edge(288)
    want = {InferenceTransforms.RESIZE_288: ref(pil.resize((tw, th), Image.BILINEAR)),
            InferenceTransforms.RESIZE_224_SQUARE: ref(pil.resize((224, 224), Image.BILINEAR))}
    th, tw = short_edge(320)
    big = pil.resize((tw, th), Image.BILINEAR)
    top, left = int(round((th - 320) / 2.0)), int(round((tw - 320) / 2.0))
    want[InferenceTransforms.RESIZE_320_CENTER] = ref(big.crop((left, top, left + 320, top + 320)))
    for t, expect in want.items():
        got = device_transform(x, t)[0].numpy()
        assert got.shape == expect.shape, (t, got.shape, expect.shape)
        step = 1.0 / 255.0 / std.min()          # one 0..255 level after Normalize
        diff = np.abs(got - expect)
        assert diff.max() <= 2.01 * step and (diff > 1.01 * step).mean() < 1e-3, (t, diff.max() / step)


class _TinyNet(torch.nn.Module):
    """Stand-in for the SSCD TorchScript file: [B, 3, H, W] -> [B, 16]."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.conv = torch.nn.Conv2d(3, 8, 5, stride=4)
        self.fc = torch.nn.Linear(8, 16)

    def forward(self, x):
        return self.fc(torch.relu(self.conv(x)).mean(dim=(2, 3)))


def _make_dataset(d, n=5):
    rng = np.random.default_rng(1)
    lens = []
    for v in range(n):
        k = int(rng.integers(2, 6))
        lens.append(k)
        np.save(d / f"Q{v:06d}.npy", rng.integers(0, 256, (k, 72, 96, 3), dtype=np.uint8))
    return lens


def test_inference_cli_torchscript_processes_and_merge(tmp_path):
    """The reference's flow (inference.py:93-158) on `.npy` frame stacks: TorchScript model file, per-rank .npz
    files, merge; 2 worker processes == 1 process; --store_fp16; externally launched ranks write their own file."""
    from vsc2022_amd.vsc.baseline import inference_cli as cli
    from vsc2022_amd.vsc.storage import load_features

    data = tmp_path / "videos"
    data.mkdir()
    lens = _make_dataset(data)
    model_path = str(tmp_path / "tiny.torchscript.pt")
    torch.jit.script(_TinyNet()).save(model_path)
    base = ["--torchscript_path", model_path, "--dataset_path", str(data), "--video_extensions", "npy",
            "--video_reader", "NPY", "--transforms", "RESIZE_224_SQUARE", "--batch_size", "3", "--fps", "2"]
    p = cli.build_parser()
    cli.main(p.parse_args(base + ["--output_file", str(tmp_path / "one" / "q.npz")]))
    one = load_features(str(tmp_path / "one" / "q.npz"))
    assert [v.video_id for v in one] == [f"Q{v:06d}" for v in range(5)]
    assert [len(v) for v in one] == lens and one[0].feature.shape[1] == 16 and one[0].feature.dtype == np.float32
    assert np.allclose(one[1].timestamps, np.stack([np.arange(lens[1]) / 2.0, (np.arange(lens[1]) + 1) / 2.0], 1))
    # eager model on the same transform == the CLI's descriptors
    frames = torch.from_numpy(np.load(data / "Q000002.npy")).permute(0, 3, 1, 2)
    with torch.no_grad():
        direct = _TinyNet()(cli.device_transform(frames, cli.InferenceTransforms.RESIZE_224_SQUARE)).numpy()
    assert np.allclose(direct, one[2].feature, atol=1e-5)
    # two spawned workers + merge: the same videos (order: rank 0's, then rank 1's), the same descriptors
    cli.main(p.parse_args(base + ["--processes", "2", "--scratch_path", str(tmp_path / "scratch"),
                                  "--output_file", str(tmp_path / "two" / "q.npz")]))
    two = {v.video_id: v for v in load_features(str(tmp_path / "two" / "q.npz"))}
    assert sorted(two) == [v.video_id for v in one]
    assert sorted(os.listdir(tmp_path / "scratch")) == ["0.npz", "1.npz"]
    assert [v.video_id for v in load_features(str(tmp_path / "scratch" / "1.npz"))] == ["Q000001", "Q000003"]
    for v in one:
        assert np.array_equal(two[v.video_id].feature, v.feature) and np.array_equal(two[v.video_id].timestamps, v.timestamps)
    # an externally launched rank writes its own share to --output_file; fp16 storage
    cli.main(p.parse_args(base + ["--distributed_rank", "2", "--distributed_size", "3", "--store_fp16",
                                  "--output_file", str(tmp_path / "r2.npz")]))
    r2 = load_features(str(tmp_path / "r2.npz"))
    assert [v.video_id for v in r2] == ["Q000002"]
    assert np.allclose(r2[0].feature, one[2].feature, atol=2e-3, rtol=2e-3)
    with pytest.raises(Exception):
        cli.main(p.parse_args(base + ["--processes", "2", "--distributed_size", "2", "--output_file", str(tmp_path / "x.npz")]))
    with pytest.raises(FileNotFoundError):  # no ffmpeg binary here: the FFMPEG reader must say so, not return nothing
        cli.main(p.parse_args(["--dataset_path", str(data), "--video_extensions", "npy", "--ffmpeg_path",
                               "/nonexistent/ffmpeg", "--torchscript_path", model_path,
                               "--output_file", str(tmp_path / "y.npz")]))


@pytest.mark.gpu
def test_inference_cli_on_the_gpu(gpu, tmp_path):
    """--accelerator cuda: the SSCD-shaped network (scripted to a TorchScript file, loaded by the CLI) on the
    device transform RESIZE_320_CENTER; descriptors equal the eager network's."""
    from vsc2022_amd.vsc.baseline import inference_cli as cli
    from vsc2022_amd.vsc.baseline.inference import build_sscd_model
    from vsc2022_amd.vsc.storage import load_features

    data = tmp_path / "videos"
    data.mkdir()
    lens = _make_dataset(data, n=3)
    model = build_sscd_model(device="cuda", channels_last=False)
    path = str(tmp_path / "sscd_random.torchscript.pt")
    torch.jit.trace(model, torch.zeros(2, 3, 320, 320, device="cuda")).save(path)
    cli.main(cli.build_parser().parse_args(
        ["--torchscript_path", path, "--accelerator", "cuda", "--dataset_path", str(data), "--video_extensions", "npy",
         "--video_reader", "NPY", "--output_file", str(tmp_path / "q.npz")]))
    vfs = load_features(str(tmp_path / "q.npz"))
    assert [len(v) for v in vfs] == lens and vfs[0].feature.shape[1] == 512
    frames = torch.from_numpy(np.load(data / "Q000001.npy")).permute(0, 3, 1, 2).cuda()
    with torch.no_grad():
        direct = model(cli.device_transform(frames, cli.InferenceTransforms.RESIZE_320_CENTER)).cpu().numpy()
    assert np.allclose(direct, vfs[1].feature, rtol=1e-3, atol=1e-4)


def _torchvision_like(model):
    """The SSCD torchvision export's layout (adapt_sscd_model.py:64-70: backbone / pool / project, torchvision's
    Bottleneck with `downsample`), holding `model`'s weights: other names, same parameter order."""
    import collections

    import torch.nn as nn

    class TVBottleneck(nn.Module):
        def __init__(self, b):
            super().__init__()
            self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3 = b.conv1, b.bn1, b.conv2, b.bn2, b.conv3, b.bn3
            self.relu = nn.ReLU()
            self.downsample = b.down

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            return self.relu(self.bn3(self.conv3(out)) + idt)

    class GeM(nn.Module):
        def forward(self, x):
            return x.clamp(min=1e-6).pow(3.0).mean(dim=(2, 3)).pow(1.0 / 3.0)

    blocks = [TVBottleneck(b) for b in model.trunk]
    backbone = nn.Sequential(collections.OrderedDict(
        conv1=model.stem[0], bn1=model.stem[1], relu=nn.ReLU(), maxpool=model.stem[3],
        layer1=nn.Sequential(*blocks[:3]), layer2=nn.Sequential(*blocks[3:7]), layer3=nn.Sequential(*blocks[7:13]),
        layer4=nn.Sequential(*blocks[13:])))
    return nn.Sequential(collections.OrderedDict(backbone=backbone, pool=GeM(), project=model.embed)).eval()


def test_sscd_weights_are_recovered_from_a_torchscript_export(tmp_path):
    """`sscd_from_module`: a traced torchvision-style export (other parameter names) -> SSCDModel with the same function;
    a different architecture -> None."""
    from vsc2022_amd.vsc.baseline.inference import build_sscd_model, sscd_from_module

    model = build_sscd_model(dims=64, seed=5, device="cpu", channels_last=False)
    g = torch.Generator().manual_seed(1)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    tv = _torchvision_like(model)
    assert list(tv.state_dict())[0] == "backbone.conv1.weight" and "backbone.layer2.0.downsample.1.running_var" in tv.state_dict()
    path = str(tmp_path / "tv.torchscript.pt")
    torch.jit.trace(tv, torch.zeros(2, 3, 64, 64)).save(path)
    loaded = torch.jit.load(path)
    back = sscd_from_module(loaded)
    assert back is not None
    x = torch.randn((3, 3, 96, 96), generator=g)
    with torch.no_grad():
        assert torch.allclose(back(x), loaded(x), rtol=1e-4, atol=1e-5) and torch.allclose(back(x), model(x), rtol=1e-4, atol=1e-5)
    assert sscd_from_module(torch.jit.script(_TinyNet())) is None
    # same shapes, another function (a BatchNorm's statistics swapped): refused by the check on a random batch
    tv.backbone.bn1.running_var.mul_(4.0)
    wrong = sscd_from_module(tv)
    with torch.no_grad():
        assert wrong is None or torch.allclose(wrong(x), tv(x), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_inference_cli_fast_flag_on_the_gpu(gpu, tmp_path):
    """`--fast` (extension): the CLI converts the TorchScript export back into SSCDModel and runs FastSSCD; descriptors
    within cosine 0.999 of the plain CLI run on every frame; a non-ResNet model is refused."""
    from vsc2022_amd.vsc.baseline import inference_cli as cli
    from vsc2022_amd.vsc.baseline.inference import build_sscd_model
    from vsc2022_amd.vsc.storage import load_features

    data = tmp_path / "videos"
    data.mkdir()
    _make_dataset(data, n=4)
    model = build_sscd_model(device="cuda", channels_last=False)
    for blk in model.trunk:
        blk.bn3.weight.fill_(0.25)
    path = str(tmp_path / "tv.torchscript.pt")
    torch.jit.trace(_torchvision_like(model), torch.zeros(2, 3, 320, 320, device="cuda")).save(path)
    base = ["--torchscript_path", path, "--accelerator", "cuda", "--dataset_path", str(data), "--video_extensions", "npy",
            "--video_reader", "NPY"]
    cli.main(cli.build_parser().parse_args(base + ["--output_file", str(tmp_path / "slow.npz")]))
    cli.main(cli.build_parser().parse_args(base + ["--fast", "--output_file", str(tmp_path / "fast.npz")]))
    slow, fast = load_features(str(tmp_path / "slow.npz")), load_features(str(tmp_path / "fast.npz"))
    assert [v.video_id for v in slow] == [v.video_id for v in fast]
    for a, b in zip(slow, fast):
        assert np.array_equal(a.timestamps, b.timestamps) and a.feature.shape == b.feature.shape
        cos = (a.feature * b.feature).sum(1) / np.linalg.norm(a.feature, axis=1) / np.linalg.norm(b.feature, axis=1)
        assert cos.min() >= 0.999, cos.min()
    tiny = str(tmp_path / "tiny.torchscript.pt")
    torch.jit.script(_TinyNet()).save(tiny)
    with pytest.raises(Exception, match="--fast"):
        cli.main(cli.build_parser().parse_args(["--torchscript_path", tiny, "--fast"] + base[2:] + ["--output_file", str(tmp_path / "x.npz")]))


def test_batchnorm_folding_keeps_the_function():
    from vsc2022_amd.vsc.baseline.inference import build_sscd_model, fold_batchnorm

    model = build_sscd_model(dims=64, seed=2, device="cpu", channels_last=False)
    # non-trivial BN statistics
    g = torch.Generator().manual_seed(0)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    fused = fold_batchnorm(model)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in fused.modules())
    x = torch.randn(3, 3, 96, 96, generator=g)
    with torch.no_grad():
        a, b = model(x), fused(x)
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), (a - b).abs().max()


@pytest.mark.gpu
def test_bias_act_epilogue_kernel_is_one_exact_rounding(gpu):
    """`vsc_bias_act_bf16` (csrc/eltwise.hip): y = act(y + bias (+ res)) on bf16 matrices, fp32 arithmetic in the order
    (y + bias) + res, one round-to-nearest-even: bit-identical to the same expression in torch, NaN / inf kept."""
    from vsc2022_amd.vsc.baseline.inference import _bias_act

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for rows, cols in ((1, 8), (37, 64), (4099, 256), (100000, 72)):
        y = (torch.randn((rows, cols), generator=g, device=dev) * 3).to(torch.bfloat16)
        r = (torch.randn((rows, cols), generator=g, device=dev) * 3).to(torch.bfloat16)
        b = torch.randn(cols, generator=g, device=dev)
        if rows > 30:
            y[5, 3], y[6, 1], y[7, 2], r[8, 0] = float("nan"), float("inf"), float("-inf"), float("nan")
            y[9, 4], b[5] = 3.0e38, 3.0e38                                          # overflows to +inf in bf16
        for res in (None, r):
            for relu in (False, True):
                want = y.float() + b
                if res is not None:
                    want = want + res.float()
                if relu:
                    want = torch.relu(want)
                want = want.to(torch.bfloat16)
                got = _bias_act(y.clone(), b, res, relu)
                assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (rows, cols, res is not None, relu)
    with pytest.raises(ValueError):
        _bias_act(torch.zeros((4, 12), device=dev, dtype=torch.bfloat16), torch.zeros(12, device=dev), None, True)


@pytest.mark.gpu
def test_fused_stem_pool_kernel_is_bit_identical_to_the_separate_passes(gpu):
    """`vsc_pool3x3s2_bias_relu_bf16` == nn.MaxPool2d(3, 2, 1)(relu(x + bias) rounded to bf16), bit for bit, odd and even
    sizes, NaN propagated."""
    from vsc2022_amd.vsc.baseline.inference import _pool_bias_relu

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    pool = torch.nn.MaxPool2d(3, stride=2, padding=1)
    for n, c, h, w in ((1, 8, 1, 1), (2, 64, 7, 9), (3, 16, 10, 6), (4, 64, 160, 160)):
        x = (torch.randn((n, c, h, w), generator=g, device=dev) * 2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if h > 5:
            x[0, 3, 2, 3] = float("nan")
            x[1, 1, 0, 0] = float("inf")
        b = torch.randn(c, generator=g, device=dev)
        want = pool(torch.relu(x.float() + b.view(1, -1, 1, 1)).to(torch.bfloat16))
        got = _pool_bias_relu(x, b)
        assert got.shape == want.shape
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        assert torch.equal(torch.nan_to_num(got.float(), nan=7.0), torch.nan_to_num(want.float(), nan=7.0)), (n, c, h, w)


@pytest.mark.gpu
def test_fast_sscd_on_other_frame_sizes(gpu):
    """FastSSCD against the fp32 eager network on frame sizes other than 320 x 320 (the CLI's 224-square transform, odd
    and non-square sizes whose feature maps end inside a 64-pixel tile, a single frame): cosine >= 0.999 per frame."""
    from vsc2022_amd.vsc.baseline.inference import FastSSCD, build_sscd_model, preprocess

    dev = torch.device("cuda", 0)
    model = build_sscd_model(device=dev)
    for blk in model.trunk:
        blk.bn3.weight.fill_(0.25)
    fast = FastSSCD(model).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    for b, h, w in ((3, 224, 224), (2, 250, 190), (1, 64, 64), (5, 97, 131), (1, 320, 320)):
        base = torch.rand((b, 3, 5, 5), generator=g, device=dev)
        frames = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear") + 0.05 * torch.rand((b, 3, h, w), generator=g, device=dev)
        x = preprocess((frames.clamp(0, 1) * 255).to(torch.uint8))
        with torch.no_grad():
            want, got = model(x).float(), fast(x).float()
        assert got.shape == want.shape == (b, 512) and torch.isfinite(got).all()
        cos = torch.nn.functional.cosine_similarity(want, got, dim=1)
        assert cos.min().item() >= 0.999, (b, h, w, cos.min().item())


@pytest.mark.gpu
def test_implicit_gemm_convolution_kernel_against_fp64(gpu):
    """`vsc_conv_bias_act_bf16` (csrc/conv_gemm.hip): 3x3 / padding 1 / stride 1 and 2 and 1x1 convolutions with bias,
    identity and ReLU inside, against the same expression in fp64 on the same bf16 values: within one bf16 rounding;
    image sizes that are odd, smaller than the kernel, pixel counts that end inside a 64-pixel tile, both channel-block
    widths (Cout = 64 / 128k), taps that fall outside the image on every side; invalid arguments are refused."""
    import torch.nn.functional as F

    from vsc2022_amd.vsc.baseline.inference import _conv1x1_rows, _conv_bias_act

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(12)

    def cl(t):
        return t.contiguous(memory_format=torch.channels_last)

    cases = [(2, 64, 5, 7, 64, 3, 1), (3, 128, 9, 6, 128, 3, 2), (1, 64, 1, 1, 64, 3, 1), (1, 64, 2, 2, 128, 3, 2), (5, 192, 17, 13, 64, 3, 2),
             (2, 64, 40, 40, 256, 3, 1), (7, 128, 21, 20, 192, 3, 1), (2, 64, 12, 12, 256, 1, 1), (2, 256, 8, 8, 512, 1, 1)]
    for b, c, h, w, n, k, stride in cases:
        x = cl(torch.randn((b, c, h, w), generator=g, device=dev).to(torch.bfloat16))
        wt = cl((torch.randn((n, c, k, k), generator=g, device=dev) / (k * k * c) ** 0.5).to(torch.bfloat16))
        bias = torch.randn(n, generator=g, device=dev)
        base = F.conv2d(x.double(), wt.double(), bias.double(), stride, k // 2)
        res = cl(torch.randn(base.shape, generator=g, device=dev).to(torch.bfloat16))
        for r in (None, res):
            for relu in (False, True):
                want = base if r is None else base + r.double()
                want = want.relu() if relu else want
                got = _conv_bias_act(x, wt, bias, r, stride, relu)
                assert tuple(got.shape) == tuple(want.shape) and got.permute(0, 2, 3, 1).is_contiguous()
                bad = (got.double() - want).abs() > want.abs() * 2.0 ** -8 + 1e-4 * (k * k * c) ** 0.5
                assert not bool(bad.any()), (b, c, h, w, n, k, stride, r is not None, relu, int(bad.sum()))
    # the [M, K] view used for the 1x1 convolutions of the trunk
    a = torch.randn((70001, 256), generator=g, device=dev).to(torch.bfloat16)
    w2 = (torch.randn((512, 256), generator=g, device=dev) / 16).to(torch.bfloat16)
    bias = torch.randn(512, generator=g, device=dev)
    r2 = torch.randn((70001, 512), generator=g, device=dev).to(torch.bfloat16)
    got = _conv1x1_rows(a, w2, bias, r2, True).double()
    want = (a.double() @ w2.double().t() + bias.double() + r2.double()).relu()
    assert not bool(((got - want).abs() > want.abs() * 2.0 ** -8 + 2e-3).any())
    with pytest.raises(ValueError):
        _conv_bias_act(cl(torch.zeros((1, 32, 4, 4), device=dev, dtype=torch.bfloat16)), cl(torch.zeros((64, 32, 3, 3), device=dev, dtype=torch.bfloat16)),
                       torch.zeros(64, device=dev), None, 1, True)
    with pytest.raises(ValueError):
        _conv_bias_act(cl(torch.zeros((1, 64, 4, 4), device=dev, dtype=torch.bfloat16)), cl(torch.zeros((64, 64, 3, 3), device=dev, dtype=torch.bfloat16)),
                       torch.zeros(64, device=dev), None, 3, True)


@pytest.mark.gpu
def test_fused_1x1_convolution_kernel_against_fp64(gpu):
    """`vsc_gemm_bias_act_bf16` (csrc/gemm_epi.hip): act(a @ w.T + bias (+ res)) with bf16 operands, fp32 accumulation and
    one rounding: every output within one bf16 rounding (2^-8 relative, + the fp32 accumulation's noise) of the same
    expression in fp64, for row counts that end inside a 64-row tile, every wave arrangement (N = 64, 128, 256, 512),
    with / without identity and ReLU; invalid shapes are refused."""
    from vsc2022_amd.vsc.baseline.inference import _gemm_bias_act

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    for M, K, N in ((1, 64, 64), (63, 64, 128), (64, 128, 256), (65, 64, 512), (1000, 128, 64), (4133, 192, 256), (70001, 64, 256)):
        a = (torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16)
        w = (torch.randn((N, K), generator=g, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
        for r in (None, res):
            for relu in (False, True):
                got = _gemm_bias_act(a, w, bias, r, relu).double()
                want = a.double() @ w.double().t() + bias.double()
                if r is not None:
                    want = want + r.double()
                if relu:
                    want = want.relu()
                tol = want.abs() * 2.0 ** -8 + 1e-4 * (K ** 0.5)
                bad = (got - want).abs() > tol
                assert not bool(bad.any()), (M, K, N, r is not None, relu, int(bad.sum()))
    # the accumulation is exact where it can be: small integers
    a = torch.randint(-4, 5, (200, 64), generator=g, device=dev).to(torch.bfloat16)
    w = torch.randint(-4, 5, (64, 64), generator=g, device=dev).to(torch.bfloat16)
    zero = torch.zeros(64, device=dev)
    assert torch.equal(_gemm_bias_act(a, w, zero, None, False).float(), (a.float() @ w.float().t()).to(torch.bfloat16).float())
    for M, K, N in ((10, 32, 64), (10, 64, 96)):
        with pytest.raises(ValueError):
            _gemm_bias_act(torch.zeros((M, K), device=dev, dtype=torch.bfloat16), torch.zeros((N, K), device=dev, dtype=torch.bfloat16),
                           torch.zeros(N, device=dev), None, True)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["fast_sscd", "autocast_folded"])
def test_fast_inference_configuration_against_fp32_eager(gpu, config):
    """Accuracy gate of the configurations that profiles/r0*_config3_inference.md TIME -- `FastSSCD` (round 3: trunk in
    bf16, 1x1 convolutions as GEMMs, fused epilogues) and round 2's bf16 autocast over the folded network; frames of
    consecutive videos packed into batches of 256, channels-last, BatchNorms folded into the convolutions -- against
    what the reference runs (vsc/baseline/inference_impl.py:210-239: fp32, eager, one video per batch), on 256
    synthetic videos x 25 frames of structured content (low-frequency patterns: iid noise frames would all map to
    one descriptor).  Stated tolerance: cosine >= 0.999 for every frame, and every frame's nearest neighbour among
    the fp32 descriptors of ALL frames is the frame itself (identical top-1 retrieval), searched on the engine."""
    from dataclasses import dataclass

    from vsc2022_amd.vsc.baseline.inference import FastSSCD, SyntheticVideos, build_sscd_model, fold_batchnorm, \
        run_inference, run_inference_packed, to_flat
    from vsc2022_amd.vsc.index import FlatIndex

    @dataclass
    class PatternVideos(SyntheticVideos):
        def video(self, idx, n_frames, device):
            g = torch.Generator(device=device)
            g.manual_seed(self.seed * 1000003 + idx)
            base = torch.rand((1, 3, 6, 6), generator=g, device=device)           # the video's scene
            frames = base + 0.35 * torch.rand((n_frames, 3, 6, 6), generator=g, device=device)
            frames = torch.nn.functional.interpolate(frames, size=(self.size, self.size), mode="bilinear")
            frames = frames + 0.03 * torch.rand(frames.shape, generator=g, device=device)
            return (frames / frames.amax(dim=(1, 2, 3), keepdim=True) * 255.0).to(torch.uint8)

    dev = torch.device("cuda", 0)
    src = PatternVideos(n_videos=256, frames=(25, 25), size=320, seed=11)
    model = build_sscd_model(device=dev)
    # A random-init trunk is a poor stand-in for trained weights in two opposite ways (measured on the MI355X,
    # scripts/dbg_infer.py): with identity BatchNorms every input collapses onto one direction (different frames
    # 0.9998 alike: retrieval meaningless), with data-calibrated BatchNorms and unit residual gains it is chaotic
    # (bf16 rounding amplified to cosine 0.989).  Trained ResNets sit in between; so does this one: BatchNorm
    # statistics taken from the data (one cumulative-average pass over 8 videos in train mode) and the last
    # BatchNorm of every residual branch scaled to 0.25, the usual small-residual initialisation.
    from vsc2022_amd.vsc.baseline.inference import preprocess

    for blk in model.trunk:
        blk.bn3.weight.fill_(0.25)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
            mod.momentum = None
    model.train()
    with torch.no_grad():
        for v in range(8):
            model(preprocess(src.video(1000 + v, 25, dev)))
    model.eval()
    slow, off, ids = to_flat(run_inference(model, src, dev, batch_size=32, autocast_dtype=None))
    if config == "fast_sscd":
        fast, off2, ids2 = to_flat(run_inference_packed(FastSSCD(model).to(dev), src, dev, batch_size=256))
    else:
        fast, off2, ids2 = to_flat(run_inference_packed(fold_batchnorm(model), src, dev, batch_size=256,
                                                        autocast_dtype=torch.bfloat16))
    assert ids == ids2 and np.array_equal(off, off2) and slow.shape == fast.shape == (256 * 25, 512)
    assert torch.isfinite(slow).all() and torch.isfinite(fast).all()
    cos = torch.nn.functional.cosine_similarity(slow, fast, dim=1)
    # descriptors of different frames must be distinguishable for the gate to mean anything
    sn = slow / slow.norm(dim=1, keepdim=True)
    spread = (sn[:2000] @ sn[2000:4000].T).max().item()
    print(f"spread {spread:.5f}  min cosine {cos.min().item():.6f}  mean cosine {cos.mean().item():.6f}")
    assert spread < 0.999, f"degenerate descriptors: different frames are {spread:.5f} alike"
    assert cos.min().item() >= 0.999, f"min cosine {cos.min().item():.5f} (mean {cos.mean().item():.5f})"
    index = FlatIndex(512)
    index.add(sn)
    fn = fast / fast.norm(dim=1, keepdim=True)
    _, top1 = index.search(fn, 1)
    assert np.array_equal(top1[:, 0], np.arange(len(fn))), f"{int((top1[:, 0] != np.arange(len(fn))).sum())} frames retrieve another frame"


def test_checked_fast_falls_back_on_a_bad_first_batch():
    """`--fast` is gated on REAL data: the first batch also runs on the fp32 network; a fast network that disagrees
    (any frame below the stated cosine) is dropped for the rest of the run."""
    from vsc2022_amd.vsc.baseline.inference_cli import CheckedFast

    torch.manual_seed(0)
    eager = torch.nn.Linear(16, 8)
    good = torch.nn.Linear(16, 8)
    good.load_state_dict(eager.state_dict())
    with torch.no_grad():
        good.weight.add_(1e-5 * torch.randn_like(good.weight))
    bad = torch.nn.Linear(16, 8)
    x = torch.randn(12, 16)
    ok = CheckedFast(good, eager, 0.999)
    y = ok(x)
    assert ok.use_fast and ok.first_batch_cosine >= 0.999 and torch.equal(y, good(x))
    assert torch.equal(ok(x), good(x))                       # later batches: the fast network, no second check
    gate = CheckedFast(bad, eager, 0.999)
    y = gate(x)
    assert not gate.use_fast and gate.first_batch_cosine < 0.999
    assert torch.equal(y, eager(x)) and torch.equal(gate(x), eager(x))
