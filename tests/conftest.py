import os
import sys

import pytest

try:  # torch bundles its own libamdhip64.so.7; importing it FIRST makes libwam_hip.so bind to the same HIP runtime
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-audio-mastering_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import golden_util
    return golden_util.load()
