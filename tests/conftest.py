import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ws():
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "watershed_ref.npz"))
    cases = {}
    for k in z.files:
        if k.startswith("neighbour_order"):
            continue
        name, field = k.split("/")
        cases.setdefault(name, {})[field] = z[k]
    return cases
