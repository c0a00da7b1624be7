import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):           # plain `pytest tests/...` from anywhere, not only `python -m pytest` from the root
    if _p not in sys.path:
        sys.path.insert(0, _p)
# Should the process ever die under a test (round 4: a GPU queue error, whose message pytest's capture of fd 2 swallowed --
# tests/test_gpu_tilemerge.py::test_first_filter_launch_on_fresh_streams), the library writes the aborting thread's native
# stack to a descriptor it is told (kmd_api.hip, abort_trace): the one pytest keeps of the real stderr (pytest_configure).

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if "KMD_ABORT_TRACE" not in os.environ:
        fd = 2
        try:                                     # the saved real stderr of the global capture (private to pytest: best effort)
            cap = config.pluginmanager.getplugin("capturemanager")._global_capturing
            if cap is not None and cap.err is not None and hasattr(cap.err, "targetfd_save"):
                fd = os.dup(cap.err.targetfd_save)
                os.set_inheritable(fd, False)
        except Exception:
            fd = 2
        os.environ["KMD_ABORT_TRACE"] = "fd:%d" % fd


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
