"""CPU: pins the oracle (oracle/liboracle.so) to the golden vectors generated from the
reference's exact CPU code (tests/golden/make_golden.py) and to independent dense algebra."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err

INV = golden_files("inv_")
SN = golden_files("selfnorm_")


def _id(p):
    return os.path.basename(p)[:-4]


def test_fixture_inventory():
    assert len(INV) >= 18 and len(SN) >= 3


@pytest.mark.parametrize("path", INV, ids=_id)
def test_inverse_matches_reference(oracle, path):
    g = load_golden(path)
    x64, w64 = g["x"].astype(np.float64), g["w"].astype(np.float64)
    z = oracle.inverse(x64, w64, g["diag"], g["order"])
    # reference inverse_op_cython.inverse_conv (fp64)
    assert rel_err(z, g["z_cython_f64"]) < 1e-12
    if "z_parallel_f64" in g:  # reference solve_parallel_mc (serial build)
        assert rel_err(z, g["z_parallel_f64"]) < 1e-12
    if "z_solve_f32" in g:  # reference solve_mc.solve, fp32 torch loops
        z32 = oracle.inverse(g["x"], g["w"], g["diag"], g["order"])
        assert z32.dtype == np.float32
        assert rel_err(z32, g["z_solve_f32"]) < 2e-6
        assert rel_err(z32, g["z_cython_f64"]) < 1e-5


@pytest.mark.parametrize("path", INV, ids=_id)
def test_forward_logdet_match_reference(oracle, path):
    g = load_golden(path)
    B, C, H, W, K = g["shape"]
    w64 = g["w"].astype(np.float64)
    xh = oracle.forward(g["z_cython_f64"], w64, g["diag"], g["order"])
    assert rel_err(xh, g["xhat_f64"]) < 1e-12
    assert rel_err(xh, g["x"]) < 1e-6  # round trip x -> z -> x (test_layers.py:19-36, atol 1e-3)
    ld = oracle.logdet(w64, H, W, g["diag"], g["order"])
    assert abs(ld - g["logdet_formula"]) < 1e-9
    if "logdet_slogdet" in g:
        assert abs(ld - g["logdet_slogdet"]) < 1e-8 * max(1.0, abs(g["logdet_slogdet"]))
    if not g["diag"]:
        assert ld == 0.0


@pytest.mark.parametrize("path", [p for p in INV if "dx_f64" in np.load(p).files], ids=_id)
def test_gradients_match_autograd(oracle, path):
    g = load_golden(path)
    B, C, H, W, K = g["shape"]
    w64 = g["w"].astype(np.float64)
    u = oracle.dy(g["g"].astype(np.float64), w64, g["diag"], g["order"])
    assert rel_err(u, g["dx_f64"]) < 1e-11
    dw = oracle.dw(g["z_cython_f64"], u, (K, K), g["diag"], g["order"])
    assert rel_err(dw, g["dw_f64"]) < 1e-11
    m = oracle.mask(C, K, K, g["diag"], g["order"], np.float64)
    assert np.array_equal(m, g["mask"])
    assert np.all(dw[m == 0] == 0)


@pytest.mark.parametrize("order", ["TL", "TR", "BL", "BR"])
@pytest.mark.parametrize("diag", [0, 1])
def test_dense_algebra(oracle, order, diag):
    rng = np.random.default_rng(5)
    B, C, H, W, K = 2, 3, 4, 5, 3
    x = rng.standard_normal((B, C, H, W))
    g = rng.standard_normal((B, C, H, W))
    w = oracle._flip(rng.standard_normal((C, C, K, K)) * 0.3, "TL")
    if diag:
        wt = w.copy()
        for c in range(C):
            wt[c, c, -1, -1] = 1.0 + 0.2 * rng.standard_normal()
        w = oracle._flip(wt, order)
    A = oracle.dense_operator(w, H, W, diag, order)
    n = C * H * W
    z = oracle.inverse(x, w, diag, order)
    assert rel_err(z, np.linalg.solve(A, x.reshape(B, n).T).T.reshape(x.shape)) < 1e-11
    assert rel_err(oracle.forward(z, w, diag, order), x) < 1e-11
    u = oracle.dy(g, w, diag, order)
    assert rel_err(u, np.linalg.solve(A.T, g.reshape(B, n).T).T.reshape(x.shape)) < 1e-11
    assert abs(oracle.logdet(w, H, W, diag, order) - np.linalg.slogdet(A)[1]) < 1e-10
    # adjoint identity <A^-T g, x> = <g, A^-1 x>
    assert abs((u * x).sum() - (g * z).sum()) < 1e-9 * abs((g * z).sum())
    # finite-difference check of dW on a few entries
    dw = oracle.dw(z, u, (K, K), diag, order)
    m = oracle.mask(C, K, K, diag, order, np.float64)
    eps = 1e-6
    for idx in [(0, 1, 0, 0), (2, 0, 1, 2), (1, 1, 0, 1), (2, 1, K - 1, K - 1), (0, 0, K - 1, K - 1)]:
        wp, wm = w.copy(), w.copy()
        wp[idx] += eps
        wm[idx] -= eps
        fd = ((oracle.inverse(x, wp, diag, order) - oracle.inverse(x, wm, diag, order)) * g).sum() / (2 * eps)
        assert abs(fd - dw[idx]) < 1e-6 * max(1.0, abs(fd)), (idx, fd, dw[idx], m[idx])


@pytest.mark.parametrize("path", SN, ids=_id)
def test_selfnorm_matches_torch_grad(oracle, path):
    g = load_golden(path)
    p = (g["pad"], g["pad"])
    bias = g.get("bias")
    z = oracle.conv2d(g["x"], g["w"], bias, p)
    assert rel_err(z, g["z"]) < 1e-12
    ig, wf, bg, wi = oracle.selfnorm_grads(g["x"], g["w"], bias, g["r"], g["gz"], p)
    assert rel_err(ig, g["dx"]) < 1e-12
    assert rel_err(wf, g["dw_fwd"]) < 1e-12
    assert rel_err(wi, g["dw_inv"]) < 1e-12
    if bias is not None:
        assert rel_err(bg, g["dbias"]) < 1e-12


def test_f32_f64_threads_agree(oracle):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 8, 6, 6)).astype(np.float32)
    w = (rng.standard_normal((8, 8, 3, 3)) * 0.05).astype(np.float32)
    z1 = oracle.inverse(x, w, 0, "TL", nthreads=1)
    z4 = oracle.inverse(x, w, 0, "TL", nthreads=4)
    assert np.array_equal(z1, z4)
    assert rel_err(z1, oracle.inverse(x.astype(np.float64), w.astype(np.float64))) < 1e-6
