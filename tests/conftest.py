import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def _release_device_scratch_between_modules(request):
    """After a module of GPU tests: the library's parked scratch (tens of GB behind the partition-size tests) goes back
    to the driver (kmd_release_cache), so that every module starts from an empty cache -- as a fresh process would."""
    yield
    if not any(m.name == "gpu" for m in request.node.iter_markers()) and "gpu" not in str(getattr(request.module, "pytestmark", "")):
        return
    try:
        import kmdiff_amd as K
        if K.device_count() >= 1:
            K._native.lib().kmd_stream_sync(None)
            K._native.lib().kmd_release_cache()
    except Exception:
        pass
