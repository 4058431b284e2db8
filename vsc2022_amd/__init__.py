"""vsc2022_amd -- MI355X (gfx950) engine for the descriptor-search / candidate /
temporal-localisation hot path of facebookresearch/vsc2022.

Layout
    csrc/ + libvscmi.so   hand-written HIP kernels behind the C ABI of include/vscmi.h
    vsc/                  host-side mirror of the reference's vsc.index / vsc.candidates /
                          vsc.baseline.{score_normalization,localization} (+ storage, metrics)
    vcsl/vta.py           build_vta_model("TN").forward_sim on the GPU
    dist.py               query-sharded multi-GPU search (one process per GPU, RCCL)
    synth.py              seeded synthetic descriptors

`vsc2022_amd.install()` registers the mirrors under the reference's module names (`vsc.index`,
`vsc.candidates`, `vcsl.vta`, ...) so that descriptor_eval.py / matching_eval.py-style drivers
import them unchanged.  There is no CPU fallback anywhere: without libvscmi.so and a gfx950
device every operation raises.
"""
import importlib
import sys

__version__ = "0.1.0"

_ALIASES = {
    "vsc": "vsc2022_amd.vsc",
    "vsc.index": "vsc2022_amd.vsc.index",
    "vsc.candidates": "vsc2022_amd.vsc.candidates",
    "vsc.metrics": "vsc2022_amd.vsc.metrics",
    "vsc.storage": "vsc2022_amd.vsc.storage",
    "vsc.descriptor_eval_lib": "vsc2022_amd.vsc.descriptor_eval_lib",
    "vsc.baseline": "vsc2022_amd.vsc.baseline",
    "vsc.baseline.score_normalization": "vsc2022_amd.vsc.baseline.score_normalization",
    "vsc.baseline.localization": "vsc2022_amd.vsc.baseline.localization",
    "vsc.baseline.sscd_baseline": "vsc2022_amd.vsc.baseline.sscd_baseline",
    "vsc.baseline.dns_baseline": "vsc2022_amd.vsc.baseline.dns_baseline",
    "vsc.baseline.dns_index": "vsc2022_amd.vsc.baseline.dns_index",
    "vcsl": "vsc2022_amd.vcsl",
    "vcsl.vta": "vsc2022_amd.vcsl.vta",
}


def install(force: bool = False) -> None:
    """Make `import vsc.index`, `import vsc.candidates`, `from vcsl.vta import build_vta_model`
    (the reference's module names) resolve to this package."""
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force:
            mod = sys.modules[alias]
            if not getattr(mod, "__name__", "").startswith("vsc2022_amd"):
                raise RuntimeError(
                    f"module {alias!r} is already imported from {getattr(mod, '__file__', '?')}; "
                    "call vsc2022_amd.install() before importing the reference, or pass force=True"
                )
            continue
        sys.modules[alias] = importlib.import_module(target)
