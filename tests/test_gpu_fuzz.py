"""Short soak run of scripts/fuzz_prefilter.py: random shapes / norms / ties, the pre-filtered routes forced on
against the all-fp32 route on the same GPU (top-K, k-NN, range search bit-identical).  A longer run of the same
script is what found the re-scoring kernel walking unwritten candidates after a candidate-list overflow."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_prefilter_vs_fp32_route(seed):
    r = subprocess.run([sys.executable, "scripts/fuzz_prefilter.py", "--seconds", "8", "--seed", str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "fuzz ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [13])
def test_fuzz_int8_prefilter_vs_fp32_route(seed):
    """The same soak with the int8 kernel on every pre-filtered batch (VSC_I8=2): dims up to 1000, rows of wildly
    different norms, exact ties, coordinates on which all references agree."""
    r = subprocess.run([sys.executable, "scripts/fuzz_prefilter.py", "--seconds", "10", "--seed", str(seed)], cwd=ROOT,
                       env=dict(os.environ, VSC_I8="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "fuzz ok" in r.stdout


@pytest.mark.gpu
def test_fuzz_pipeline_vs_oracle():
    """Random small datasets through candidates + TN localisation, every output equal to the CPU oracle."""
    r = subprocess.run([sys.executable, "scripts/fuzz_pipeline.py", "--seconds", "8", "--seed", "21"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "fuzz ok" in r.stdout
