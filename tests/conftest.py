import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "inverse-flow_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    d = np.load(path, allow_pickle=False)
    out = {k: d[k] for k in d.files}
    for k in ("order",):
        if k in out:
            out[k] = str(out[k])
    for k in ("diag", "pad"):
        if k in out:
            out[k] = int(out[k])
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    n = np.linalg.norm(b.ravel())
    d = np.linalg.norm((a - b).ravel())
    return d / n if n > 0 else d


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
