import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def no_stale_thread_error(request):
    """Every GPU test must leave the calling thread's device-less error slot (rtcGetDeviceError(NULL), first-error-wins, cleared on
    read: kernels/common/device.cpp:273-330) empty: a stale code would be what the NEXT test reads.  Only for tests that run on the GPU box."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from embree_amd import api as A
    L = A.load()
    L.rtcGetDeviceError(None)                                # whatever an earlier, failed test left behind is not this test's fault
    yield
    left = L.rtcGetDeviceError(None)
    assert left == 0, "test left RTCError %d in the thread's device-less error slot" % left
