import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    if not os.path.exists(os.path.join(ge.PKG_DIR, "lib", "libmf.so")):
        ge.build()
    p = ge.import_package()
    if os.environ.get("MFX_TEST_LIB"):  # an experiment build of the library (make variant) under the same tests
        p.LIB_PATH = os.environ["MFX_TEST_LIB"]
    return p


@pytest.fixture(scope="session")
def orc():
    return ge.import_oracle()


@pytest.fixture(scope="session")
def toy():
    return np.load(os.path.join(GOLDEN, "toy.npz"))


@pytest.fixture(scope="session")
def small():
    return np.load(os.path.join(GOLDEN, "small.npz"))


def unique_pairs(rng, m, n, nnz, node):
    idx = rng.choice(m * n, nnz, replace=False)
    R = np.zeros(nnz, dtype=node)
    R["u"], R["v"] = idx // n, idx % n
    R["r"] = (rng.integers(2, 11, nnz) * 0.5).astype(np.float32)
    return R
