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
    """Native libraries are built in-tree.  Every library is rebuilt when the CONTENT of its sources differs
    from what it was built from (digest beside the .so, rust_raytracer_amd/build.py), so a stale prebuilt
    library can never be what the tests run — here or on the GPU box (hipcc is in the image there too)."""
    from rust_raytracer_amd import build as b
    b.build_host()
    b.build_device()
    b.build_cli()
    b.build_oracle()
    b.build_tools()
    yield


@pytest.fixture(scope="session")
def repo_dir():
    return REPO


def load_scene(args):
    from rust_raytracer_amd import api
    return api.HostScene(list(args))
