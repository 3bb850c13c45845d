import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces must exist; build them once if the tree is fresh."""
    need = [os.path.join(ROOT, "flye_amd", "lib", "libflyegpu.so"),
            os.path.join(ROOT, "flye_amd", "lib", "libflyesynth.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def golden_cases():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "cases.json")) as f:
        return json.load(f)
