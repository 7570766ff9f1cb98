"""CPU: the C oracle against the compiled reference modules in oracle/_ref (built by
`make -C oracle` from the reference's .pyx files where they lie; skipped if absent)."""
import glob
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err

REFDIR = os.path.join(ROOT, "oracle", "_ref")
pytestmark = pytest.mark.skipif(not glob.glob(os.path.join(REFDIR, "inverse_op_cython*.so")),
                                reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REFDIR)
    import inverse_op_cython
    import solve_parallel_mc
    return inverse_op_cython, solve_parallel_mc


@pytest.mark.parametrize("shape", [(2, 3, 7, 7, 3), (1, 16, 9, 12, 3), (2, 64, 6, 6, 3), (3, 2, 5, 5, 2)])
def test_random_shapes(oracle, ref, shape):
    inv_cy, sp = ref
    B, C, H, W, K = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, C, H, W))
    w = rng.standard_normal((C, C, K, K)) * 0.05
    z = oracle.inverse(x, w, 0, "TL", nthreads=2)
    we = oracle.effective_weight(w, 0, "TL")
    assert rel_err(z, inv_cy.inverse_conv(x.copy(), we)) < 1e-12
    if 2 * W - 1 >= H + W - 1:
        assert rel_err(z, np.asarray(sp.solve_parallel(x.copy(), w.copy(), (K, K)))) < 1e-12
    # general diagonal: inverse_op_cython divides by the diagonal
    wd = w.copy()
    for c in range(C):
        wd[c, c, -1, -1] = 1.0 + 0.2 * rng.standard_normal()
    zd = oracle.inverse(x, wd, 1, "TL")
    assert rel_err(zd, inv_cy.inverse_conv(x.copy(), oracle.effective_weight(wd, 1, "TL"))) < 1e-12
