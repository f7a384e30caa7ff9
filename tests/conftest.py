import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def models():
    return {s: json.load(open(os.path.join(GOLDEN, "models_%s.json" % s))) for s in ("0.1", "0.25", "0.5")}


@pytest.fixture(scope="session")
def aer_counts():
    return {s: json.load(open(os.path.join(GOLDEN, "aer_counts_%s.json" % s))) for s in ("0.1", "0.25", "0.5")}


@pytest.fixture(scope="session")
def config1():
    return json.load(open(os.path.join(GOLDEN, "config1.json")))


def random_theta(dim, scale=0.5, seed=1984):
    """theta law of /root/reference/run_experiment.py:3,30"""
    from scipy.stats import halfnorm
    rs = np.random.RandomState(seed)
    return (-halfnorm.rvs(loc=0, scale=scale, size=dim, random_state=rs)).tolist()
