import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["c1_er1000", "rr50_d4", "rr200_s64", "two_triangles", "two_hexagons", "d16_er2000", "d4_rr300"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU case")


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)
