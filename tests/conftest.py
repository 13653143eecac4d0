import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


collect_ignore = ["golden/ref_suite"]      # the reference's own tests: run by test_gpu_ref_suite.py against the drop-in


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product library (hipcc cross-compiles without a GPU) and the CPU oracle."""
    from pyahocorasick_amd.build import build_libacx
    build_libacx(verbose=False)
    from oracle import orc
    orc.lib()
    yield


@pytest.fixture(scope="session")
def have_gpu():
    import pyahocorasick_amd as A
    return A.device_count() > 0


@pytest.fixture(autouse=True)
def _host_walk_policy(request):
    """The walk over the host trie (include/acx.h §4b) is for BASELINE config 1, processes without a device and haystacks that
    do not pay a launch.  No GPU parity result may come from it: every `-m gpu` test runs with the walk switched off (limit -1,
    also in the environment of the subprocesses that import the drop-in module) and fails if the library's counter of host
    walks moved while it ran.  Every other test runs with the library's default."""
    from pyahocorasick_amd import _lib
    l = _lib.lib()
    gpu = request.node.get_closest_marker("gpu") is not None
    old_env = os.environ.get("ACX_HOST_WALK_BYTES")
    if gpu:
        l.acx_set_host_walk_bytes(-1)
        os.environ["ACX_HOST_WALK_BYTES"] = "-1"
    else:
        l.acx_set_host_walk_bytes(2048)
        os.environ.pop("ACX_HOST_WALK_BYTES", None)
    before = l.acx_host_walk_calls()
    yield
    after = l.acx_host_walk_calls()
    l.acx_set_host_walk_bytes(2048)
    if old_env is None:
        os.environ.pop("ACX_HOST_WALK_BYTES", None)
    else:
        os.environ["ACX_HOST_WALK_BYTES"] = old_env
    if gpu:
        assert after == before, "a GPU test was answered by the host walk (%d calls)" % (after - before)
