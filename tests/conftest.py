import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


def _gpu_count():
    try:
        from vsc2022_amd import _lib

        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """Fail loudly (not skip) when a GPU test runs without the HIP extension or a device."""
    from vsc2022_amd import _lib

    n = _lib.device_count()
    assert n > 0, "no gfx950 device visible: GPU tests need a real MI355X (there is no CPU fallback)"
    return n


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.build()
    return oracle
