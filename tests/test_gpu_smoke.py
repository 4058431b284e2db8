"""The driver's smoke() in a FRESH process.

A fresh process has no neighbouring allocations, so an out-of-bounds access of a device buffer
faults instead of landing in some other test's memory (this is how the walk past the hit arrays
after a kept-hit overflow was found; csrc/select.hip select_begin_kernel).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_in_fresh_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout


@pytest.mark.gpu
def test_parity_suites_with_poisoned_allocations():
    """VSC_POISON_ALLOC=1 fills every fresh device buffer with 0xFF (NaN scores, -1 indices): a kernel
    that consumes memory nobody wrote fails the bit-exact parity tests instead of passing on the
    zero-filled pages a fresh allocation usually has."""
    env = dict(os.environ, VSC_POISON_ALLOC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_search.py",
                        "tests/test_gpu_tn.py", "tests/test_gpu_edge_cases.py", "tests/test_gpu_golden.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_overflow_retry_in_fresh_process():
    """First search of the process overflows the default kept-hit buffer and is rerun with a larger
    one: results must still be the exact top-K prefix."""
    code = r"""
import sys
sys.path[:0] = [%r, %r]
import numpy as np
import oracle as orc
from vsc2022_amd.vsc.index import FlatIndex
rng = np.random.default_rng(5)
R = rng.standard_normal((900, 64)).astype(np.float32)
Q = rng.standard_normal((300, 64)).astype(np.float32)
idx = FlatIndex(64); idx.add(R)
# 32 rows x 900 refs = 2K exactly: the first batch does not re-threshold, the second one (64 rows)
# pushes the kept list to 86400 > the default buffer of 4K + 1024 = 58624 entries
K = 14400
i, j, s, rad = idx.global_topk(Q, K)
oi, oj, os_ = orc.global_threshold_search(Q, R, K)
assert np.array_equal(i, oi) and np.array_equal(j, oj)
assert np.array_equal(s.view(np.uint32), os_.view(np.uint32))
print("overflow ok", len(s))
""" % (ROOT, os.path.join(ROOT, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "overflow ok" in r.stdout
