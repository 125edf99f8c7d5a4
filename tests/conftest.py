import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment(request):
    """libsober_hip reads its A/B / test switches of the environment once, at load time.  A test that flips one with
    `monkeypatch` tells the library (sober_amd._native.reload_switches); after EVERY gpu test the library reads the
    restored environment again, so that no switch outlives the test that set it."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import sober_amd._native as nat
        if nat._lib is not None:
            nat.reload_switches()
