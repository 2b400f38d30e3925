import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_artifacts():
    """The libraries are build products (kept out of git, shipped with the tree to the GPU box).
    A clean checkout has none: build them once, as __graft_entry__.build() does."""
    need = [os.path.join(ROOT, "raytracer_amd", "librthip.so"), os.path.join(ROOT, "raytracer_amd", "rt_headless"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def rt():
    import raytracer_amd
    return raytracer_amd
