"""ctypes binding of libvscmi.so (the C ABI declared in include/vscmi.h).

The shared library is the product; this module only loads it, declares the prototypes and turns
VSC_ERR_* codes into Python exceptions (the reference's convention is exceptions/asserts:
vsc/index.py:37-40, vsc/storage.py:49-57, vsc/baseline/localization.py:59,64).

There is NO CPU fallback: if the library is missing, or no gfx950 device is visible when an
operation is issued, the call raises.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSCMI_LIB") or os.path.join(_HERE, "libvscmi.so")  # VSCMI_LIB: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")

VSC_OK = 0
VSC_ERR_INVALID = -1
VSC_ERR_HIP = -2
VSC_ERR_NOMEM = -3
VSC_ERR_CAPACITY = -4
VSC_ERR_OVERFLOW = -5
VSC_ERR_NODEVICE = -6
MEM_HOST = 0
MEM_DEVICE = 1
METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
TN_MAX_BOXES = 16

# every symbol include/vscmi.h declares
EXPORTS = (
    "vsc_last_error", "vsc_version", "vsc_device_count",
    "vsc_index_create", "vsc_index_destroy", "vsc_index_add", "vsc_index_ntotal", "vsc_index_dim",
    "vsc_index_metric", "vsc_index_set_hit_capacity", "vsc_index_sync", "vsc_index_knn",
    "vsc_index_set_option", "vsc_index_get_option", "vsc_index_set_stream", "vsc_tn_set_stream", "vsc_set_aux_stream",
    "vsc_index_range_search", "vsc_index_global_topk", "vsc_index_global_topk_seeded", "vsc_index_candidates", "vsc_pair_max", "vsc_sort_hits", "vsc_row_normalize",
    "vsc_score_histogram", "vsc_score_pick", "vsc_filter_hits", "vsc_argsort_scores", "vsc_merge_topk",
    "vsc_tn_create", "vsc_tn_set_queries", "vsc_tn_destroy", "vsc_tn_localize", "vsc_tn_forward_sim", "vsc_tn_similarity",
    "vsc_index_profile", "vsc_index_profile_read", "vsc_index_profile_read_class", "vsc_index_search_stats",
    "vsc_aux_profile", "vsc_aux_profile_read", "vsc_bias_act_bf16", "vsc_gemm_bias_act_bf16", "vsc_pool3x3s2_bias_relu_bf16", "vsc_conv_bias_act_bf16",
)


class TNParams(ctypes.Structure):
    _fields_ = [
        ("tn_max_step", ctypes.c_int32),
        ("tn_top_k", ctypes.c_int32),
        ("max_path", ctypes.c_int32),
        ("min_length", ctypes.c_int32),
        ("min_sim", ctypes.c_float),
        ("max_iou", ctypes.c_float),
    ]


class VscError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libvscmi error {code}: {message}")
        self.code = code


_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile libvscmi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "vscmi.h"))
    if (
        not force
        and os.path.exists(LIB_PATH)
        and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)
    ):
        return LIB_PATH
    subprocess.check_call(["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)])
    if os.environ.get("VSC_SKIP_LINT") != "1":
        # a fresh library is checked for the hazards the compiler cannot see inside the kernels' inline assembly
        # (scripts/lint_isa.py: VALU-written SGPRs read by VMEM too early, registers of outstanding stream loads touched
        # before their s_waitcnt -- what a different register allocation could silently introduce)
        # ADVISORY here: the library compiled and stays usable whatever the lint says -- a missing llvm-objdump or a
        # false positive must not take the product down; `make lint` and tests/test_isa_lint.py are the hard gate
        import sys

        lint = os.path.join(os.path.dirname(_HERE), "scripts", "lint_isa.py")
        if os.path.exists(lint):
            try:
                res = subprocess.run([sys.executable, lint, LIB_PATH], capture_output=True, text=True)
                if res.returncode != 0:
                    print(f"[vsc2022_amd] ISA lint of {LIB_PATH} reported problems (advisory; `make -C {CSRC} lint` is the "
                          f"gate):\n{res.stdout[-2000:]}{res.stderr[-2000:]}", file=sys.stderr)
            except OSError as exc:
                print(f"[vsc2022_amd] ISA lint could not run ({exc}); the library is built", file=sys.stderr)
    return LIB_PATH


def _preload_torch_hip_runtime():
    """One HIP/HSA runtime per process.

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm).
    If libvscmi pulled in the system copies first and torch its bundled ones later, two HSA runtimes
    would fight over the device ("no ROCm-capable device is detected").  Loading torch's copies
    first (without importing torch) makes both users share them, whatever the import order.
    """
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        pi64 = ctypes.POINTER(ctypes.c_int64)
        pf32 = ctypes.POINTER(ctypes.c_float)
        pi32 = ctypes.POINTER(ctypes.c_int32)
        L.vsc_last_error.restype = ctypes.c_char_p
        L.vsc_version.restype = i32
        L.vsc_device_count.restype = i32
        L.vsc_index_create.argtypes = [i32, i32, i32, ctypes.POINTER(vp)]
        L.vsc_index_destroy.argtypes = [vp]
        L.vsc_index_add.argtypes = [vp, vp, i64, i32]
        L.vsc_index_ntotal.argtypes = [vp]
        L.vsc_index_ntotal.restype = i64
        L.vsc_index_dim.argtypes = [vp]
        L.vsc_index_metric.argtypes = [vp]
        L.vsc_index_set_hit_capacity.argtypes = [vp, i64]
        L.vsc_index_sync.argtypes = [vp]
        L.vsc_index_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
        L.vsc_index_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.vsc_index_set_stream.argtypes = [vp, vp, i32]
        L.vsc_tn_set_stream.argtypes = [vp, vp, i32]
        L.vsc_set_aux_stream.argtypes = [i32, vp, i32]
        L.vsc_index_knn.argtypes = [vp, vp, i64, i32, i32, vp, vp, i32]
        L.vsc_index_range_search.argtypes = [vp, vp, i64, i32, f32, vp, vp, vp, i64, pi64]
        L.vsc_index_global_topk.argtypes = [vp, vp, i64, i32, i64, vp, vp, vp, i64, i32, pi64, pf32]
        L.vsc_index_global_topk_seeded.argtypes = [vp, vp, i64, i32, i64, f32, vp, vp, vp, i64, i32, pi64, pf32]
        L.vsc_index_candidates.argtypes = [vp, vp, i64, i32, i64, vp, vp, vp, vp, vp, i64, pi64, pi64]
        L.vsc_pair_max.argtypes = [vp, vp, vp, i64, i32, vp, i64, vp, i64, i32, vp, vp, vp, vp, i64,
                                   i32, pi64, i32]
        L.vsc_sort_hits.argtypes = [vp, vp, vp, i64, i32, i64, i64, vp, vp, vp, i32, i32]
        L.vsc_row_normalize.argtypes = [vp, i64, i32, i32, vp, i32, i32]
        L.vsc_score_histogram.argtypes = [vp, i64, vp, i32, vp, i32]
        L.vsc_score_pick.argtypes = [vp, vp, i32, i32]
        L.vsc_filter_hits.argtypes = [vp, vp, vp, i64, f32, vp, vp, vp, pi64, i32]
        L.vsc_argsort_scores.argtypes = [vp, i64, i32, vp, i32, i32]
        L.vsc_merge_topk.argtypes = [vp, vp, i64, i32, i32, vp, vp, i32]
        L.vsc_tn_create.argtypes = [vp, vp, i64, vp, vp, i64, i32, i32, i32, ctypes.POINTER(vp)]
        L.vsc_tn_set_queries.argtypes = [vp, vp, vp, i64, i32]
        L.vsc_tn_destroy.argtypes = [vp]
        L.vsc_tn_localize.argtypes = [vp, vp, vp, i64, i32, ctypes.POINTER(TNParams), f32, vp, vp, vp, i32]
        L.vsc_tn_forward_sim.argtypes = [vp, vp, vp, vp, i64, ctypes.POINTER(TNParams), vp, vp, vp, i32]
        L.vsc_tn_similarity.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, f32, vp, i64, pi32, pi32]
        L.vsc_index_profile.argtypes = [vp, i32]
        L.vsc_index_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), pi64,
                                             ctypes.POINTER(ctypes.c_double), i32]
        L.vsc_index_profile_read_class.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), pi64,
                                                   ctypes.POINTER(ctypes.c_double), i32]
        L.vsc_index_search_stats.argtypes = [vp, pi64, pi64]
        L.vsc_aux_profile.argtypes = [i32]
        L.vsc_aux_profile_read.argtypes = [i32, ctypes.POINTER(ctypes.c_double), pi64, ctypes.POINTER(ctypes.c_double), i32]
        L.vsc_bias_act_bf16.argtypes = [vp, vp, vp, i64, i64, i32, vp]
        L.vsc_gemm_bias_act_bf16.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, vp]
        L.vsc_pool3x3s2_bias_relu_bf16.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp]
        L.vsc_conv_bias_act_bf16.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, i32, vp]
        for name in EXPORTS:
            fn = getattr(L, name)
            if fn.restype is ctypes.c_int and name not in ("vsc_version", "vsc_device_count"):
                fn.restype = ctypes.c_int
        _lib = L
        return _lib


def check(rc: int):
    if rc == VSC_OK:
        return
    msg = lib().vsc_last_error().decode("utf-8", "replace")
    if rc == VSC_ERR_INVALID:
        raise ValueError(f"libvscmi: {msg}")
    raise VscError(rc, msg)


def device_count() -> int:
    return int(lib().vsc_device_count())


def default_device() -> int:
    """One process per GPU: LOCAL_RANK picks the device (torch.distributed launch convention)."""
    n = device_count()
    if n <= 0:
        raise RuntimeError(
            "libvscmi: no gfx950 (MI355X) device is visible; this engine has no CPU fallback"
        )
    return int(os.environ.get("LOCAL_RANK", "0")) % n


def ptr(a):
    """(address, mem kind) of a numpy array (host) or a torch tensor (host or HBM)."""
    if a is None:
        return None, MEM_HOST
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return a.ctypes.data, MEM_HOST
    # torch tensor (duck-typed: no torch import needed here)
    assert a.is_contiguous(), "tensor must be contiguous"
    return a.data_ptr(), (MEM_DEVICE if a.is_cuda else MEM_HOST)


def f32c(x):
    """fp32 C-contiguous numpy view/copy (inputs may be fp16/float64: vsc/index.py a-1)."""
    return np.ascontiguousarray(x, dtype=np.float32)
