import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "timed-design_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# the load-time guard keeps its verdicts in a per-user directory (csrc/runtime.hip guard_disk_path); the suite must measure, not
# remember: no verdict files unless a test points TH_GUARD_CACHE at a directory of its own
os.environ.setdefault("TH_GUARD_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from timed_hip import _lib
        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand in the build container; prebuilt on the GPU box)."""
    from timed_hip import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_lib()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(lib):
    n = _gpu_count()
    if n < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return 0


@pytest.fixture(scope="session")
def cnn_golden():
    import json
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "cnn_golden.npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(scope="session")
def sampler_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden.npz"))
