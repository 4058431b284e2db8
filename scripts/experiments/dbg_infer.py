import sys; sys.path.insert(0,'.')
import numpy as np, torch
from dataclasses import dataclass
from vsc2022_amd.vsc.baseline.inference import SyntheticVideos, build_sscd_model, fold_batchnorm, run_inference, run_inference_packed, to_flat, preprocess
@dataclass
class PatternVideos(SyntheticVideos):
    def video(self, idx, n_frames, device):
        g = torch.Generator(device=device); g.manual_seed(self.seed * 1000003 + idx)
        base = torch.rand((1, 3, 6, 6), generator=g, device=device)
        frames = base + 0.35 * torch.rand((n_frames, 3, 6, 6), generator=g, device=device)
        frames = torch.nn.functional.interpolate(frames, size=(self.size, self.size), mode="bilinear")
        frames = frames + 0.03 * torch.rand(frames.shape, generator=g, device=device)
        return (frames / frames.amax(dim=(1, 2, 3), keepdim=True) * 255.0).to(torch.uint8)
dev = torch.device("cuda", 0)
src = PatternVideos(n_videos=64, frames=(25, 25), size=320, seed=11)
for gamma in (1.0, 0.5, 0.25, 0.1, 0.0):
  for calib in (True, False):
    model = build_sscd_model(device=dev)
    for blk in model.trunk: blk.bn3.weight.fill_(gamma)
    if calib:
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d): mod.reset_running_stats(); mod.momentum = None
        model.train()
        with torch.no_grad():
            for v in range(8): model(preprocess(src.video(1000 + v, 25, dev)))
        model.eval()
    slow,_,_ = to_flat(run_inference(model, src, dev, batch_size=32, autocast_dtype=None))
    fast,_,_ = to_flat(run_inference_packed(fold_batchnorm(model), src, dev, batch_size=256, autocast_dtype=torch.bfloat16))
    cos = torch.nn.functional.cosine_similarity(slow, fast, dim=1)
    sn = slow / slow.norm(dim=1, keepdim=True); fn = fast / fast.norm(dim=1, keepdim=True)
    S = sn @ sn.T; S.fill_diagonal_(-1)
    top1 = (fn @ sn.T).argmax(1)
    mism = int((top1 != torch.arange(len(fn), device=dev)).sum())
    # margin: self-similarity minus best other
    margin = ((fn*sn).sum(1) - (fn @ sn.T + torch.diag(torch.full((len(fn),), -9.0, device=dev))).max(1).values)
    print(f"gamma {gamma} calib {calib}: nearest-other max {S.max().item():.5f} mean {S.mean().item():.4f} | cos min {cos.min().item():.6f} mean {cos.mean().item():.6f} | top1 mismatches {mism}/{len(fn)} min margin {margin.min().item():.5f} finite {bool(torch.isfinite(slow).all())}", flush=True)
