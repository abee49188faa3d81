import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Native libraries are built in-tree; on the GPU box the prebuilt .so files are used as-is
    (the build step is a no-op when they are newer than their sources)."""
    from rust_raytracer_amd import build as b
    b.build_host()
    b.build_oracle()
    if not os.path.exists(os.path.join(REPO, "rust_raytracer_amd", "librt_mi355.so")):
        b.build_device()
    b.build_tools()
    yield


@pytest.fixture(scope="session")
def repo_dir():
    return REPO


def load_scene(args):
    from rust_raytracer_amd import api
    return api.HostScene(list(args))
