import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_sessionstart(session):
    """Build the HIP library if the checkout has none (a fresh clone: *.so is git-ignored).  Building is not a
    fallback: without hipcc, or if the build fails, the product still refuses to run (SageLibraryError)."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "sageattention_amd", "libsage_gfx950.so")
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if not os.path.exists(lib) and hipcc:
        subprocess.call(["make", "-C", os.path.join(ROOT, "sageattention_amd", "csrc"), "-j", "8", "-s"])


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(autouse=True)
def _fresh_prepass_guard(request):
    """GPU tests start with the one-launch pre-pass allowed on every device: a test that makes a pre-pass give up (on purpose, or because it
    crowds the chip) trips the per-device guard, which would silently move every later test of the process onto the kernel sequence."""
    if request.node.get_closest_marker("gpu") is not None:
        from sageattention_amd import quant
        for g in quant._PrepassGuard._by_device.values():
            g.reset()
    yield
