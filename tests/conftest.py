import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session")
def built():
    """Native pieces are built once per session (no-op when up to date / prebuilt)."""
    from strelka_amd import build as sk_build
    if os.path.exists("/opt/rocm/bin/hipcc"):
        sk_build.build_all(verbose=False)
    from oracle import pyoracle
    pyoracle.build(quiet=True)
    return True


@pytest.fixture(scope="session")
def gpu(built):
    from strelka_amd import capi
    capi.init(0)  # raises with the library's own message when no gfx950 device is present
    yield capi
    capi.shutdown()
