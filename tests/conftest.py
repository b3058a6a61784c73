import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG_NAME = "av-simulation-at-intersections_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (hyphenated directory name -> importlib)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference path (test infrastructure, oracle/)."""
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def routes(pkg):
    """Synthetic route table, yaw-smoothed like MPC.__init__ does (main/lib/mpc.py:260)."""
    rs = pkg.synth.make_route_table()
    for r in rs:
        pkg.synth.smooth_yaw_inplace(r[:, 2])
    return rs


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
