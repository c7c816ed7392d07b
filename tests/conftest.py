import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".json") and not name.startswith("input_"):
        with open(path) as f:
            return json.load(f)
    with open(path) as f:
        return f.read()


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def entry():
    import __graft_entry__ as g
    return g


@pytest.fixture(scope="session")
def oracle(entry):
    """(c_oracle, poseidon2_ref): the CPU checker.  Test infrastructure only."""
    c, p = entry.load_oracle()
    c.build()
    return c, p


@pytest.fixture(scope="session")
def pkg(entry):
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def ctx(pkg):
    """A cp2_ctx on GPU 0.  No fallback: on a box without a usable gfx950 device this raises."""
    c = pkg.Context(0)
    yield c
    c.close()
