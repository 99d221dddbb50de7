import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import nmfk_oracle

    nmfk_oracle.build()
    return nmfk_oracle


@pytest.fixture(scope="session")
def bss_X():
    import numpy as np

    return np.loadtxt(os.path.join(ROOT, "tests", "golden", "bss_notebook_X.txt"))
