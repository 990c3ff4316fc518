import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Builds the oracle (always) and the HIP library when it is missing (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    lib = os.path.join(ROOT, 'real_robots_amd', 'csrc', 'librealrobot_hip.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'real_robots_amd', 'csrc')])
    yield


def pytest_collection_modifyitems(config, items):
    """A machine without an AMD GPU device node cannot run the `gpu` tests (the library has no CPU fallback: rr_create fails with
    RR_EDEVICE): they are reported as skipped there, not as errors.  On a GPU box nothing is skipped."""
    if os.path.exists('/dev/kfd'):
        return
    skip = pytest.mark.skip(reason="no AMD GPU device node (/dev/kfd) on this machine; the HIP path has no CPU fallback")
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
