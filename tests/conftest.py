import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "alphasnake-zero_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_state(z, i, prefix="st_"):
    """compact state dict number i of a tic_/corner golden file"""
    return {k: z[prefix + k][i] for k in ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")}


@pytest.fixture(scope="session")
def oracle():
    from oracle import snake_oracle
    snake_oracle.lib()
    return snake_oracle
