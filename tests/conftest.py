import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def rsaf_lib():
    """The built C-ABI library (built on demand here; prebuilt on the GPU box)."""
    from robust_speech_analysis_framework_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_library(verbose=False)
    return _lib.load()
