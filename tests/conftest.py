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
