"""pytest configuration: `gpu` marker, repo root on sys.path, shared fixtures."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mirt():
    import weekend_raytracer_wgpu_amd as m
    return m


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def gpu_ctx(mirt):
    """One Context on cuda:0 for the whole GPU session (tests run in ONE process)."""
    ctx = mirt.Context(0)
    yield ctx
    ctx.close()
