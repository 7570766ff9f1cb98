"""GPU parity tests proper: the HIP path (through the C ABI, via invflow_hip) against
  (1) the committed golden vectors generated from the reference's exact CPU code,
  (2) the CPU oracle on the same seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full size (round trip, adjoint identity).
Tolerance: fp32 path, relative L2 error <= 1e-5 vs the fp64 exact inverse (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5
ORDERS = ["TL", "TR", "BL", "BR"]
INV = golden_files("inv_")


def _id(p):
    return os.path.basename(p)[:-4]


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import invflow_hip
    invflow_hip.lib()  # fails loudly if the HIP library is missing
    return invflow_hip


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def host(t):
    return t.detach().cpu().double().numpy()


def flags_of(g, H):
    return H.FLAG_GENERAL_DIAG if g["diag"] else 0


# ---------------------------------------------------------------------------------------------
# (1) golden vectors
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", INV, ids=_id)
def test_golden_inverse_forward(H, path):
    g = load_golden(path)
    x, w = dev(g["x"]), dev(g["w"])
    fl = flags_of(g, H)
    z = H.inverse(x, w, g["order"], fl)
    assert rel_err(host(z), g["z_cython_f64"]) < TOL
    if "z_solve_f32" in g:
        assert rel_err(host(z), g["z_solve_f32"]) < TOL
    xh, ld = H.forward(dev(g["z_cython_f64"]), w, g["order"], fl, want_logdet=True)
    assert rel_err(host(xh), g["xhat_f64"]) < TOL
    assert np.allclose(host(ld), g["logdet_formula"], rtol=1e-5, atol=1e-5)
    if not g["diag"]:
        assert float(ld.abs().max()) == 0.0
    # round trip x -> z -> x, the reference's own test (tests/inf/test_layers.py:19-36, atol 1e-3)
    np.testing.assert_allclose(host(H.forward(z, w, g["order"], fl)), g["x"], atol=1e-3)


@pytest.mark.parametrize("path", [p for p in INV if "dx_f64" in np.load(p).files], ids=_id)
def test_golden_backward(H, path):
    g = load_golden(path)
    fl = flags_of(g, H)
    w = dev(g["w"])
    z = dev(g["z_cython_f64"])
    dx, dw, _ = H.backward(dev(g["g"]), z, w, g["order"], fl)
    assert rel_err(host(dx), g["dx_f64"]) < TOL
    assert rel_err(host(dw), g["dw_f64"]) < TOL
    assert np.all(host(dw)[g["mask"] == 0] == 0)
    # split entry points agree with the fused call
    dx2, _, _ = H.backward(dev(g["g"]), None, w, g["order"], fl, need_dw=False)
    assert torch.equal(dx, dx2)
    K = int(g["shape"][4])
    dw2 = H.dw_from(z, dx, (K, K), g["order"], fl)
    assert torch.equal(dw, dw2)


# ---------------------------------------------------------------------------------------------
# (2) oracle on seeded inputs
# ---------------------------------------------------------------------------------------------
def _weights(rng, C, KH, KW, kind, order, oracle, diag=0):
    if kind == "refinit":  # inf/layers/inv_conv.py:153-170
        w = np.zeros((C, C, KH, KW))
        for c in range(C):
            w[c, c, KH // 2, KW // 2] = 1.0
        w += rng.standard_normal(w.shape) * 0.01 * np.sqrt(2.0 / (2 * C * KH * KW))
        w[:, -1, -1, -1] = 1.0
    else:  # inf/layers/conv.py:67-74
        w = rng.standard_normal((C, C, KH, KW)) * float(kind)
    if diag:
        for c in range(C):
            w[c, c, -1, -1] = (1.0 + 0.2 * rng.standard_normal()) * (-1 if c % 4 == 1 else 1)
    return oracle._flip(np.ascontiguousarray(w), order)


CASES = [
    # B, C, H, W, KH, KW, kind, order, diag
    (4, 64, 16, 16, 3, 3, "refinit", "TL", 0),
    (2, 64, 32, 32, 3, 3, "refinit", "TL", 0),
    (3, 32, 32, 32, 3, 3, "0.02", "BR", 0),
    (2, 12, 16, 16, 3, 3, "0.05", "TR", 0),  # if_glow_cifar channel counts 12/24/48
    (2, 24, 8, 8, 3, 3, "0.05", "BL", 0),
    (3, 48, 8, 8, 3, 3, "0.03", "TL", 0),
    (2, 48, 4, 4, 3, 3, "0.03", "TR", 0),
    (3, 14, 7, 7, 2, 2, "0.05", "BL", 0),  # the small layers' MFMA weight gradient: rows that are no multiple of four pixels,
    (2, 13, 10, 6, 3, 3, "0.05", "BR", 1),  # channel counts that are no multiple of the tile
    (64, 1, 28, 28, 3, 3, "refinit", "TL", 0),  # config 1: if_cnn_mnist 28x28x1, batch 64
    (5, 4, 14, 14, 2, 2, "0.05", "TL", 0),  # config 3: if_glow_mnist 2x2 kernels after squeeze
    (5, 8, 7, 7, 2, 2, "0.05", "TL", 0),
    (2, 6, 9, 5, 2, 3, "0.05", "TR", 0),  # KH != KW, H != W
    (2, 6, 5, 9, 3, 2, "0.05", "BL", 1),
    (2, 16, 8, 8, 3, 3, "0.05", "TL", 1),
    (1, 3, 1, 7, 3, 3, "0.1", "TL", 0),  # single row
    (1, 3, 7, 1, 3, 3, "0.1", "BR", 0),  # single column
    (2, 5, 6, 6, 1, 1, "0.2", "TL", 0),  # 1x1 kernel: pure in-pixel triangular solve
    (1, 256, 4, 4, 3, 3, "0.01", "TL", 0),  # config 5 channel count
    (2, 7, 33, 20, 3, 3, "0.03", "TL", 0),  # ragged sizes
    (3, 64, 32, 32, 3, 3, "refinit", "TR", 0),  # MFMA scan + MFMA conv (x^ = A z), every order and tile shape
    (3, 64, 20, 32, 3, 3, "0.02", "BL", 0),     # rows not a multiple of the conv's 8-row band / the scan's tiles
    (2, 64, 16, 16, 2, 2, "0.05", "BR", 0),
    (2, 32, 32, 16, 2, 2, "0.05", "TR", 1),
    (2, 32, 8, 32, 3, 3, "0.05", "TL", 1),
    (2, 48, 32, 32, 3, 3, "0.02", "BR", 0),   # padded channels on the MFMA scan (48 -> 64, 20 -> 32, 9 -> 32, 33 -> 64)
    (2, 20, 16, 12, 2, 2, "0.05", "TR", 1),
    (1, 9, 20, 8, 3, 3, "0.05", "BL", 0),
    (2, 33, 12, 16, 3, 3, "0.03", "TL", 1),
    (3, 256, 8, 8, 3, 3, "0.01", "TR", 0),    # wide layers: one GEMM per anti-diagonal over all images (scan_wide.hip)
    (2, 128, 6, 10, 2, 2, "0.02", "BL", 1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "b%dc%d_%dx%d_k%dx%d_%s_%s_d%d" % c)
def test_against_oracle(H, oracle, case):
    B, C, Hh, Ww, KH, KW, kind, order, diag = case
    rng = np.random.default_rng(1000 + CASES.index(case))
    x = rng.standard_normal((B, C, Hh, Ww))
    g = rng.standard_normal((B, C, Hh, Ww))
    w = _weights(rng, C, KH, KW, kind, order, oracle, diag)
    fl = H.FLAG_GENERAL_DIAG if diag else 0
    x32, g32, w32 = x.astype(np.float32), g.astype(np.float32), w.astype(np.float32)
    xo, go, wo = x32.astype(np.float64), g32.astype(np.float64), w32.astype(np.float64)
    z_o = oracle.inverse(xo, wo, diag, order, nthreads=8)
    u_o = oracle.dy(go, wo, diag, order, nthreads=8)
    dw_o = oracle.dw(z_o, u_o, (KH, KW), diag, order, nthreads=8)
    xd, gd, wd = dev(x32), dev(g32), dev(w32)
    z = H.inverse(xd, wd, order, fl)
    assert rel_err(host(z), z_o) < TOL
    xh, ld = H.forward(z, wd, order, fl, want_logdet=True)
    assert rel_err(host(xh), xo) < TOL
    assert np.allclose(host(ld), oracle.logdet(wo, Hh, Ww, diag, order), rtol=1e-5, atol=1e-4)
    dx, dw, _ = H.backward(gd, z, wd, order, fl)
    assert rel_err(host(dx), u_o) < TOL
    assert rel_err(host(dw), dw_o) < TOL
    m = oracle.mask(C, KH, KW, diag, order)
    assert np.all(host(dw)[m == 0] == 0)


@pytest.mark.parametrize("shape", [(3, 64, 32, 32, 3, 1), (2, 64, 12, 16, 3, 1), (2, 32, 16, 32, 3, 1), (2, 32, 32, 16, 3, 1),
                                   (2, 64, 16, 16, 3, 0), (2, 48, 16, 16, 3, 1)],
                         ids=lambda s_: "b%dc%d_%dx%d_k%d_p%d" % s_)
def test_dense_conv_pieces(H, oracle, shape):
    """SelfNormConv's dense contractions (selfnorm.py:42-82): conv2d, backward_input, backward_weight through the C ABI
    against the oracle; same-size shapes with C in {32,64} run on the MFMA conv kernel, the others on the direct one."""
    B, C, Hh, Ww, K, p = shape
    rng = np.random.default_rng(7 + B + C + Hh)
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    w = (0.05 * rng.standard_normal((C, C, K, K))).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    y_o = oracle.conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), (p, p), nthreads=8)
    y = H.conv2d(dev(x), dev(w), dev(b), (p, p))
    assert rel_err(host(y), y_o) < TOL
    gz = rng.standard_normal(y_o.shape).astype(np.float32)
    gx_o = oracle.conv2d_igrad(gz.astype(np.float64), w.astype(np.float64), x.shape, (p, p), nthreads=8)
    gx = H.conv2d_igrad(dev(gz), dev(w), x.shape, (p, p))
    assert rel_err(host(gx), gx_o) < TOL
    gw_o = oracle.conv2d_wgrad(gz.astype(np.float64), x.astype(np.float64), w.shape, (p, p), nthreads=8)
    gw = H.conv2d_wgrad(dev(gz), dev(x), w.shape, (p, p))
    assert rel_err(host(gw), gw_o) < TOL


@pytest.mark.parametrize("amp", [0.02, 0.03, 0.05], ids=lambda a: "w%g" % a)
def test_scan_beyond_fp16_range(H, oracle, amp):
    """center-tap identity + amp*randn weights at C=64, 32x32 grow r along the sweep beyond the fp16 range (max|z| about
    1e5, 1e7 and 1e11): the MFMA scan rescales the image once (x * 2^-12) and, past that, falls back to its exact fp32
    body (scan_mfma.hip); the result stays within the tolerance of the exact solver either way, forward and adjoint."""
    rng = np.random.default_rng(21)
    B, C, Hh, Ww, K = 2, 64, 32, 32, 3
    w = np.zeros((C, C, K, K)); w[np.arange(C), np.arange(C), 1, 1] = 1.0  # torch.nn.init.dirac_: the CENTER tap
    w = (w + amp * rng.standard_normal((C, C, K, K))).astype(np.float32)
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    z_o = oracle.inverse(x.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
    u_o = oracle.dy(g.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
    assert np.all(np.isfinite(z_o)) and np.abs(z_o).max() > 6.0e4  # the case does leave the fp16 range
    z = H.inverse(dev(x), dev(w))
    assert rel_err(host(z), z_o) < TOL
    dx, dw, _ = H.backward(dev(g), z, dev(w))
    assert rel_err(host(dx), u_o) < TOL
    dw_o = oracle.dw(z_o, u_o, (K, K), 0, "TL", nthreads=8)
    # dW sums products of two tensors that each span up to 1e11 and peak in opposite corners: the weight-gradient
    # kernel's per-tensor power-of-two prescale keeps 22 bits below each tensor's maximum only (measured 3e-4 at 1e7)
    assert rel_err(host(dw), dw_o) < 2e-3


def test_dense_conv_fp16_range(H, oracle):
    """A band whose input leaves the fp16 range is redone in fp32 by the same workgroup (conv_mfma.hip): finite,
    fp32-accurate results, and the other bands keep their split-fp16 results."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 64, 32, 32)).astype(np.float32)
    x[1, 5, 13, 7] = 3.0e5  # one pixel beyond fp16: its band (rows 8..15 of image 1) takes the fp32 route
    w = (0.05 * rng.standard_normal((64, 64, 3, 3))).astype(np.float32)
    y_o = oracle.conv2d(x.astype(np.float64), w.astype(np.float64), None, (1, 1), nthreads=8)
    y = host(H.conv2d(dev(x), dev(w), None, (1, 1)))
    assert np.all(np.isfinite(y))
    assert rel_err(y, y_o) < TOL


def test_recon_term(H, oracle):
    """recon_weight * mean_b ||x - A z||^2 added to dW inside the fused backward (z deliberately
    perturbed so the residual is not zero); oracle = closed form -(2 rw/B) sum r (x) shifted z."""
    rng = np.random.default_rng(3)
    B, C, Hh, Ww, K = 3, 8, 6, 6, 3
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    w = (rng.standard_normal((C, C, K, K)) * 0.05).astype(np.float32)
    z = oracle.inverse(x.astype(np.float64), w.astype(np.float64)) + 0.01 * rng.standard_normal((B, C, Hh, Ww))
    z = z.astype(np.float32)
    rw = 0.7
    dx, dw, rl = H.backward(dev(g), dev(z), dev(w), "TL", 0, x=dev(x), recon_weight=rw)
    z64, w64 = z.astype(np.float64), w.astype(np.float64)
    u = oracle.dy(g.astype(np.float64), w64)
    r = x.astype(np.float64) - oracle.forward(z64, w64)
    dw_o = oracle.dw(z64, u + (2 * rw / B) * r, (K, K))
    assert rel_err(host(dw), dw_o) < TOL
    assert rel_err(host(dx), u) < TOL
    assert abs(float(rl) - (r ** 2).sum() / B) < 1e-4 * (r ** 2).sum() / B


@pytest.mark.parametrize("order", ["TL", "TR", "BL", "BR"])
@pytest.mark.parametrize("shape", [(4, 64, 32, 32, 3), (2, 32, 16, 16, 2)], ids=lambda s: "b%dc%d_%dx%d_k%d" % s)
def test_recon_term_mfma_route(H, oracle, shape, order):
    """The recon term on the MFMA route (x^ = A z by k_conv_mfma with the effective weight packed in the same launch, the
    mixed gradient through k_wgrad_mfma without carried maxima), every order, against the oracle's closed form."""
    B, C, Hh, Ww, K = shape
    rng = np.random.default_rng(40 + C + ORDERS.index(order))
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    w = _weights(rng, C, K, K, "refinit", order, oracle).astype(np.float32)
    w64 = w.astype(np.float64)
    z = (oracle.inverse(x.astype(np.float64), w64, 0, order, nthreads=8) + 0.01 * rng.standard_normal((B, C, Hh, Ww))).astype(np.float32)
    rw = 0.35
    dx, dw, rl = H.backward(dev(g), dev(z), dev(w), order, 0, x=dev(x), recon_weight=rw)
    z64 = z.astype(np.float64)
    u = oracle.dy(g.astype(np.float64), w64, 0, order, nthreads=8)
    r = x.astype(np.float64) - oracle.forward(z64, w64, 0, order, nthreads=8)
    dw_o = oracle.dw(z64, u + (2 * rw / B) * r, (K, K), 0, order, nthreads=8)
    assert rel_err(host(dx), u) < TOL
    assert rel_err(host(dw), dw_o) < TOL
    assert abs(float(rl) - (r ** 2).sum() / B) < 1e-4 * (r ** 2).sum() / B
    # the caller's loss word RECEIVES the loss (invflow.h): through the C ABI with the word pre-filled with garbage, twice
    L = H.lib()
    gd, zd, xd, wd = dev(g), dev(z), dev(x), dev(w)
    dx2, dw2 = torch.empty_like(gd), torch.empty_like(wd)
    nb = L.ifl_workspace_bytes(H.OP_BACKWARD, B, C, Hh, Ww, K, K, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    loss = torch.full((1,), 1.0e9, device="cuda")
    for _ in range(2):
        rc = L.ifl_backward_f32(gd.data_ptr(), zd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dx2.data_ptr(), dw2.data_ptr(), rw,
                                loss.data_ptr(), B, C, Hh, Ww, K, K, ORDERS.index(order), 0, ws.data_ptr(), nb, None,
                                H.scan_state(gd.device).data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, H.last_error() if hasattr(H, "last_error") else rc
        assert abs(float(loss) - (r ** 2).sum() / B) < 1e-4 * (r ** 2).sum() / B
    # and the reverse pass with its log-det, which shares the packing launch
    xh, ld = H.forward(dev(z), dev(w), order, 0, want_logdet=True)
    assert rel_err(host(xh), oracle.forward(z64, w64, 0, order, nthreads=8)) < TOL and float(ld.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# (3) BASELINE.json full size: B=128, C=64, 32x32, K=3
# ---------------------------------------------------------------------------------------------
def test_full_size_properties(H, oracle):
    torch.manual_seed(0)
    B, C, Hh, Ww, K = 128, 64, 32, 32, 3
    rng = np.random.default_rng(0)
    w = _weights(rng, C, K, K, "refinit", "TL", oracle).astype(np.float32)
    x = torch.randn(B, C, Hh, Ww, device="cuda")
    g = torch.randn(B, C, Hh, Ww, device="cuda")
    wd = dev(w)
    z = H.inverse(x, wd)
    xh = H.forward(z, wd)
    # encode -> decode round trip
    assert float((xh - x).norm() / x.norm()) < TOL
    dx, dw, _ = H.backward(g, z, wd)
    # adjoint identity <A^-T g, x> = <g, A^-1 x>
    lhs = float((dx.double() * x.double()).sum())
    rhs = float((g.double() * z.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float(dx.double().norm() * x.double().norm())
    # A^T dx = g
    # (checked through the forward operator of the transposed problem = linearity of dW in g below)
    # oracle on a sub-batch (fp64)
    sub = [0, 57, 127]
    z_o = oracle.inverse(host(x[sub]), w.astype(np.float64), nthreads=3)
    assert rel_err(host(z[sub]), z_o) < TOL
    u_o = oracle.dy(host(g[sub]), w.astype(np.float64), nthreads=3)
    assert rel_err(host(dx[sub]), u_o) < TOL
    # dW: linearity over the batch -- dW(full) == dW(first half) + dW(second half), and the
    # sub-batch agrees with the oracle
    h = B // 2
    dwa = H.dw_from(z[:h].contiguous(), dx[:h].contiguous(), (K, K))
    dwb = H.dw_from(z[h:].contiguous(), dx[h:].contiguous(), (K, K))
    assert float((dwa + dwb - dw).norm() / dw.norm()) < TOL
    dws = H.dw_from(z[sub].contiguous(), dx[sub].contiguous(), (K, K))
    dw_o = oracle.dw(z_o, u_o, (K, K), nthreads=8)
    assert rel_err(host(dws), dw_o) < TOL
    # determinism: same inputs, bit-identical outputs
    z2 = H.inverse(x, wd)
    dx2, dw2, _ = H.backward(g, z, wd)
    assert torch.equal(z, z2) and torch.equal(dx, dx2) and torch.equal(dw, dw2)


def test_carry_side_channel_is_transparent(H, oracle):
    """The forward -> backward carry (adjoint fold + maxima) must not change any result."""
    rng = np.random.default_rng(21)
    for (B, C, S, K, order) in [(4, 64, 32, 3, "TL"), (3, 32, 16, 2, "BR"), (2, 8, 8, 3, "TL")]:
        w = dev(_weights(rng, C, K, K, "refinit", order, oracle))
        x = torch.randn(B, C, S, S, device="cuda")
        g = torch.randn(B, C, S, S, device="cuda") * 1e-3  # small gradients: the dx prescale matters
        z0 = H.inverse(x, w, order)
        dx0, dw0, _ = H.backward(g, z0, w, order)
        carry = H.new_carry(w)
        z1 = H.inverse(x, w, order, carry=carry)
        dx1, dw1, _ = H.backward(g, z1, w, order, carry=carry)
        assert torch.equal(z0, z1) and torch.equal(dx0, dx1)
        assert float((dw0 - dw1).norm() / dw0.norm()) < 1e-6
        u_o = oracle.dy(host(g), host(w), 0, order, nthreads=8)
        dw_o = oracle.dw(host(z0), u_o, (K, K), 0, order, nthreads=8)
        assert rel_err(host(dw1), dw_o) < TOL
        # a second backward with the same carry (retain_graph) still works
        dx2, dw2, _ = H.backward(g, z1, w, order, carry=carry)
        assert torch.equal(dx1, dx2) and rel_err(host(dw2), dw_o) < TOL


def test_empty_batch_and_errors(H):
    w = torch.zeros(4, 4, 3, 3, device="cuda")
    x = torch.zeros(0, 4, 5, 5, device="cuda")
    assert H.inverse(x, w).shape == (0, 4, 5, 5)
    assert H.forward(x, w).shape == (0, 4, 5, 5)
    dx, dw, _ = H.backward(x, x, w)
    assert dx.shape == x.shape and float(dw.abs().sum()) == 0.0
    xc = torch.zeros(2, 4, 5, 5)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        H.inverse(xc, w)
    xs = torch.zeros(2, 4, 5, 10, device="cuda")[:, :, :, ::2]
    with pytest.raises(RuntimeError, match="must be contiguous"):
        H.inverse(xs, w)
    with pytest.raises(RuntimeError, match="does not match"):
        H.inverse(torch.zeros(2, 5, 5, 5, device="cuda"), w)
    with pytest.raises(ValueError, match="unknown order"):
        H.inverse(torch.zeros(2, 4, 5, 5, device="cuda"), w, "XX")
    # a shape the library cannot place in LDS fails loudly with the library's message
    with pytest.raises(RuntimeError, match="LDS"):
        H.inverse(torch.zeros(1, 256, 64, 64, device="cuda"), torch.zeros(256, 256, 3, 3, device="cuda"),
                  flags=H.FLAG_NO_MFMA)


@pytest.mark.parametrize("shape", [(128, 64, 32, 32, 3), (128, 32, 32, 32, 3), (40, 64, 24, 32, 2), (13, 32, 17, 32, 2),
                                   (128, 32, 32, 16, 3), (13, 48, 32, 32, 3), (5, 64, 17, 4, 3), (3, 32, 31, 12, 3)],
                         ids=lambda s: "B%d_C%d_%dx%d_K%d" % s)
def test_split_scan_equals_whole_image_scan(H, shape):
    """Two workgroups per image (the duo scan: mailbox hand-off of rows 14, 15; the shapes with 32 or 64 channels on
    32-pixel rows) against one workgroup per image (IFL_FLAG_WHOLE_IMAGE), at the bench's full size and on ragged ones (H
    not a multiple of 16, a batch that is not a multiple of 8; the other shapes run the same kernel either way).
    The arithmetic per tile is the same instruction sequence on the same operands, so the results are bit-identical --
    on every one of many back-to-back launches (tags are per-image launch generations: a stale granule of an earlier
    launch must never be taken for a fresh one), for all four orders, forward and adjoint."""
    B, C, Hh, Ww, K = shape
    gen = torch.Generator().manual_seed(5)
    w = torch.nn.init.dirac_(torch.empty(C, C, K, K)) + 0.02 * torch.randn(C, C, K, K, generator=gen)
    w[:, -1, -1, -1] = 1.0
    w = w.cuda()
    x = torch.randn(B, C, Hh, Ww, generator=gen).cuda()
    for order in ("TL", "TR", "BL", "BR"):
        z_ref = H.inverse(x, w, order, H.FLAG_WHOLE_IMAGE)
        dx_ref, dw_ref, _ = H.backward(x, z_ref, w, order, H.FLAG_WHOLE_IMAGE)
        for it in range(12 if order == "TL" else 2):
            z = H.inverse(x, w, order)
            assert torch.equal(z, z_ref), (order, it)
        dx, dw, _ = H.backward(x, z_ref, w, order)
        assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    torch.cuda.synchronize()


def test_split_scan_redoes_void_images_whole(H, oracle):
    """Images whose r leaves the fp16 range in the LOWER half only, in both halves, and in neither, mixed in one batch:
    the lower half's workgroup redoes a void image whole (scaled retry / fp32 body) and the flags say which."""
    rng = np.random.default_rng(33)
    B, C, Hh, Ww, K = 6, 64, 32, 32, 3
    w = np.zeros((C, C, K, K)); w[np.arange(C), np.arange(C), 1, 1] = 1.0
    w = (w + 0.03 * rng.standard_normal((C, C, K, K))).astype(np.float32)  # grows r to ~1e7 over a full image
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    x[1] *= 1e-4       # stays inside fp16 everywhere
    x[2, :, :12] = 0   # growth starts late: only the lower half overflows
    x[4] *= 1e-5
    z_o = oracle.inverse(x.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
    z = H.inverse(dev(x), dev(w))
    for b in range(B):
        assert rel_err(host(z)[b], z_o[b]) < TOL, b
    z_w = H.inverse(dev(x), dev(w), "TL", H.FLAG_WHOLE_IMAGE)
    for b in range(B):
        assert rel_err(host(z_w)[b], z_o[b]) < TOL, b


def test_split_scan_streams_and_graph_replay(H):
    """The split scan's hand-off state is per (device, stream) and its tags are device-side generations: two streams run
    split scans concurrently without sharing a mailbox, and a captured launch replays correctly any number of times
    (a per-launch host argument would be frozen into the graph)."""
    torch.manual_seed(7)
    B, C, Hh, Ww, K = 24, 64, 32, 32, 3
    w = torch.nn.init.dirac_(torch.empty(C, C, K, K)) + 0.02 * torch.randn(C, C, K, K)
    w[:, -1, -1, -1] = 1.0
    w = w.cuda()
    xs = [torch.randn(B, C, Hh, Ww, device="cuda") for _ in range(2)]
    refs = [H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE) for x in xs]
    torch.cuda.synchronize()
    # two side streams, interleaved launches
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(6):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k].append(H.inverse(xs[k], w))
    torch.cuda.synchronize()
    for k in range(2):
        for z in outs[k]:
            assert torch.equal(z, refs[k])
    # graph capture + replay on one of them (its state block exists: the launches above created it)
    s = streams[0]
    zg = torch.empty_like(xs[0])
    xin = xs[0].clone()
    with torch.cuda.stream(s):
        H.inverse(xin, w, out=zg)  # warm-up on the capture stream
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            H.inverse(xin, w, out=zg)
    for k in (0, 1, 0):
        xin.copy_(xs[k])
        zg.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(zg, refs[k]), k


def test_scan_state_is_caller_owned(H):
    """The two-workgroup scan runs only on a block the caller passes (scan_state: the library never allocates and keeps no
    registry).  With IFL_FLAG_WHOLE_IMAGE one workgroup per image sweeps the same two tiles through the same mailbox and
    computes the same bits; without a block the round-1 whole-image kernel runs (its own summation order: the same
    result within the tolerance against the exact solver, not bit for bit); a block that is not 256-byte aligned is
    refused; z aliasing x is refused."""
    torch.manual_seed(9)
    B, C, Hh, Ww, K = 8, 64, 32, 32, 3
    w = torch.nn.init.dirac_(torch.empty(C, C, K, K)) + 0.02 * torch.randn(C, C, K, K)
    w[:, -1, -1, -1] = 1.0
    w = w.cuda()
    x = torch.randn(B, C, Hh, Ww, device="cuda")
    z_split = H.inverse(x, w)  # (the host layer passes this stream's block)
    L = H.lib()
    stream = torch.cuda.current_stream().cuda_stream
    assert (torch.cuda.current_device(), stream) in H._scan_states
    nb = L.ifl_workspace_bytes(H.OP_INVERSE, B, C, Hh, Ww, K, K, 0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    z_none = torch.empty_like(x)
    assert L.ifl_inverse_f32(x.data_ptr(), w.data_ptr(), z_none.data_ptr(), B, C, Hh, Ww, K, K, 0, 0, ws.data_ptr(), nb, None,
                             None, stream) == 0
    z_whole = H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE)
    assert torch.equal(z_split, z_whole)
    assert float((z_split - z_none).norm() / z_split.norm()) < 2e-6
    own = torch.zeros(L.ifl_scan_state_bytes() + 256, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    off = (-own.data_ptr()) % 256
    z_own = torch.empty_like(x)
    for _ in range(3):
        assert L.ifl_inverse_f32(x.data_ptr(), w.data_ptr(), z_own.data_ptr(), B, C, Hh, Ww, K, K, 0, 0, ws.data_ptr(), nb, None,
                                 own.data_ptr() + off, stream) == 0
        assert torch.equal(z_own, z_split)
    assert L.ifl_inverse_f32(x.data_ptr(), w.data_ptr(), z_own.data_ptr(), B, C, Hh, Ww, K, K, 0, 0, ws.data_ptr(), nb, None,
                             own.data_ptr() + off + 16, stream) == -1  # IFL_EINVAL
    assert b"256-byte aligned" in L.ifl_last_error()
    assert L.ifl_inverse_f32(x.data_ptr(), w.data_ptr(), x.data_ptr(), B, C, Hh, Ww, K, K, 0, 0, ws.data_ptr(), nb, None, None,
                             stream) == -1
    assert b"must not alias" in L.ifl_last_error()
    with pytest.raises(RuntimeError, match="must not alias"):
        H.inverse(x, w, out=x)


# ---------------------------------------------------------------------------------------------
# wide layers (64 < C <= 256): the resident team scan, its fold and the W = 8 weight gradient (csrc/wide.hip)
# ---------------------------------------------------------------------------------------------
WIDE = [
    # B, C, H, W, K, order, diag
    (16, 256, 8, 8, 3, "TL", 0),   # BASELINE configs[4]: the C = 256 layers at the per-GPU batch
    (3, 256, 8, 8, 3, "BR", 1),
    (40, 256, 8, 8, 3, "TR", 0),   # more tiles than teams: several sweeps per team
    (5, 128, 16, 16, 3, "BL", 0),  # one image per tile
    (7, 96, 7, 5, 2, "TR", 1),     # ragged: H not a power of two, a partly filled tile
    (9, 192, 4, 8, 2, "TL", 0),    # four images per tile
    (2, 160, 3, 20, 3, "BR", 0),
]


@pytest.mark.parametrize("case", WIDE, ids=lambda c: "b%dc%d_%dx%d_k%d_%s_d%d" % c)
def test_wide_layers(H, oracle, case):
    B, C, Hh, Ww, K, order, diag = case
    rng = np.random.default_rng(4000 + WIDE.index(case))
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    w = _weights(rng, C, K, K, "0.01", order, oracle, diag).astype(np.float32)
    fl = H.FLAG_GENERAL_DIAG if diag else 0
    xo, go, wo = x.astype(np.float64), g.astype(np.float64), w.astype(np.float64)
    z_o = oracle.inverse(xo, wo, diag, order, nthreads=8)
    u_o = oracle.dy(go, wo, diag, order, nthreads=8)
    dw_o = oracle.dw(z_o, u_o, (K, K), diag, order, nthreads=8)
    xd, gd, wd = dev(x), dev(g), dev(w)
    for use_carry in (False, True):
        carry = H.new_carry(wd) if use_carry else None
        z = H.inverse(xd, wd, order, fl, carry=carry)
        assert rel_err(host(z), z_o) < TOL
        dx, dw, _ = H.backward(gd, z, wd, order, fl, carry=carry)
        assert rel_err(host(dx), u_o) < TOL
        assert rel_err(host(dw), dw_o) < TOL
        m = oracle.mask(C, K, K, diag, order)
        assert np.all(host(dw)[m == 0] == 0)
    # the launch-per-diagonal route (no scan state in use) gives the same answer within the tolerance
    z1 = H.inverse(xd, wd, order, fl | H.FLAG_WHOLE_IMAGE)
    assert rel_err(host(z1), z_o) < TOL
    # twice on the same stream: flags of the previous launch are stale by their generation, not by a clear
    z2 = H.inverse(xd, wd, order, fl)
    assert torch.equal(z2, z)


def test_wide_layers_beyond_fp16_range(H, oracle):
    """values beyond the fp16 range (here: the input itself) void the launch: the gated fp32 scan redoes the batch"""
    rng = np.random.default_rng(77)
    B, C, Hh, Ww, K = 4, 256, 8, 8, 3
    x = (rng.standard_normal((B, C, Hh, Ww)) * 1.0e5).astype(np.float32)
    w = _weights(rng, C, K, K, "0.01", "TL", oracle).astype(np.float32)
    z_o = oracle.inverse(x.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
    z = H.inverse(dev(x), dev(w))
    assert rel_err(host(z), z_o) < TOL
    g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    u_o = oracle.dy(g.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
    dx, dw, _ = H.backward(dev(g), z, dev(w))
    assert rel_err(host(dx), u_o) < TOL
    assert rel_err(host(dw), oracle.dw(z_o, u_o, (K, K), 0, "TL", nthreads=8)) < TOL


@pytest.mark.parametrize("pad", [(2, 2), (0, 0), (2, 0), (0, 2)], ids=lambda p: "p%d%d" % p)
def test_wide_weight_gradient_pieces(H, oracle, pad):
    """the W = 8 weight gradient as a plain convolution weight gradient, every padding corner"""
    rng = np.random.default_rng(5)
    B, C, Hh, Ww, K = 6, 128, 6, 8, 3
    x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    gz = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
    # corner pads as the layer uses them: (pt, pl) in {0, K-1}; conv2d_wgrad takes symmetric pads only, so compare through
    # the layer call: dW = -wgrad(dx, z) with the order that has this corner
    order = {(2, 2): "TL", (2, 0): "TR", (0, 2): "BL", (0, 0): "BR"}[pad]
    dw = H.dw_from(dev(x), dev(gz), (K, K), order)
    dw_o = oracle.dw(x.astype(np.float64), gz.astype(np.float64), (K, K), 0, order, nthreads=8)
    assert rel_err(host(dw), dw_o) < TOL


def test_mailbox_regions_do_not_depend_on_the_channel_count(H):
    """An image's mailbox region is the same 80 KiB whatever C is, so a granule left by a launch of another channel count
    can only carry an older tag of the SAME image.  Deterministic: three C = 64 launches of two images, then one C = 32
    launch of four on one caller-owned block; afterwards the tags found in image b's region are bounded by image b's own
    generation -- images 2 and 3, new to the block, hold nothing but the tag of their single launch -- and every result
    equals the whole-image scan bit for bit."""
    torch.manual_seed(12)
    L = H.lib()
    stream = torch.cuda.current_stream().cuda_stream
    own = torch.zeros(L.ifl_scan_state_bytes() + 256, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    off = (-own.data_ptr()) % 256

    def run(B, C):
        K, Hh, Ww = 3, 32, 32
        w = torch.nn.init.dirac_(torch.empty(C, C, K, K)) + 0.02 * torch.randn(C, C, K, K)
        w[:, -1, -1, -1] = 1.0
        w, x = w.cuda(), torch.randn(B, C, Hh, Ww, device="cuda")
        nb = L.ifl_workspace_bytes(H.OP_INVERSE, B, C, Hh, Ww, K, K, 0)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        z = torch.empty_like(x)
        assert L.ifl_inverse_f32(x.data_ptr(), w.data_ptr(), z.data_ptr(), B, C, Hh, Ww, K, K, 0, 0, ws.data_ptr(), nb, None,
                                 own.data_ptr() + off, stream) == 0
        assert torch.equal(z, H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE)), (B, C)

    for _ in range(3):
        run(2, 64)
    run(4, 32)
    torch.cuda.synchronize()
    st = own[off:off + 512 + 4 * 80 * 1024].cpu().numpy()
    gen = st[:512].view(np.uint32)
    assert list(gen[:4]) == [8, 8, 2, 2]  # two per launch
    for b in range(4):
        tags = st[512 + b * 80 * 1024:512 + (b + 1) * 80 * 1024].view(np.uint32)[1::2]
        assert tags.max() == gen[b] - 1 and (tags.max() > 0)  # this launch's tag = generation before it + 1
        if b >= 2:
            assert set(np.unique(tags)) <= {0, 1}


def test_wide_team_scan_streams_and_graph_replay(H, oracle):
    """The team scan's flags carry device-side generations of the stream's state block: two streams run team scans side by
    side, and a captured inverse + backward replays correctly any number of times."""
    torch.manual_seed(8)
    B, C, Hh, Ww, K = 6, 256, 8, 8, 3
    voided_before = H.scan_voided(torch.device("cuda"))
    rng = np.random.default_rng(8)
    w = dev(_weights(rng, C, K, K, "0.01", "TL", oracle).astype(np.float32))
    xs = [torch.randn(B, C, Hh, Ww, device="cuda") for _ in range(2)]
    refs = [H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE) for x in xs]  # (launch per diagonal: exact fp32)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(5):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k].append(H.inverse(xs[k], w))
    torch.cuda.synchronize()
    for k in range(2):
        for z in outs[k]:
            assert torch.equal(z, outs[k][0]) and rel_err(host(z), host(refs[k])) < TOL
    s = streams[0]
    xin, gin = xs[0].clone(), torch.randn_like(xs[0])
    zg, dxg, dwg = torch.empty_like(xin), torch.empty_like(xin), torch.empty_like(w)
    carry = H.new_carry(w)
    with torch.cuda.stream(s):
        H.inverse(xin, w, out=zg, carry=carry)
        H.backward(gin, zg, w, dx_out=dxg, dw_out=dwg, carry=carry)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            H.inverse(xin, w, out=zg, carry=carry)
            H.backward(gin, zg, w, dx_out=dxg, dw_out=dwg, carry=carry)
    for k in (0, 1, 0):
        xin.copy_(xs[k])
        zg.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(zg, outs[k][0]), k
        dx_e, dw_e, _ = H.backward(gin, zg, w)
        assert torch.equal(dxg, dx_e) and rel_err(host(dwg), host(dw_e)) < 1e-6
    assert H.scan_voided(xin.device) == voided_before


@pytest.mark.parametrize("Hh", [1, 2, 3, 5, 15, 16, 17, 18, 19, 31])
def test_duo_scan_row_counts(H, oracle, Hh):
    """the duo scan at every kind of row count -- fewer rows than the first burst of row loads, one full tile, a lower tile
    of one to three rows, a ragged lower tile -- for both channel widths, both kernel sizes and reflected orders: equal to the
    whole-image kernel bit for bit, and to the oracle within the tolerance (forward and adjoint)"""
    rng = np.random.default_rng(600 + Hh)
    for (C, K, order) in ((64, 3, "TL"), (32, 2, "BR"), (64, 2, "TR"), (32, 3, "BL")):
        B, Ww = 3, 32
        x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
        g = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
        w = _weights(rng, C, K, K, "0.02", order, oracle).astype(np.float32)
        xd, gd, wd = dev(x), dev(g), dev(w)
        z = H.inverse(xd, wd, order)
        assert torch.equal(z, H.inverse(xd, wd, order, H.FLAG_WHOLE_IMAGE)), (Hh, C, K, order)
        z_o = oracle.inverse(x.astype(np.float64), w.astype(np.float64), 0, order, nthreads=8)
        assert rel_err(host(z), z_o) < TOL, (Hh, C, K, order)
        dx, dw, _ = H.backward(gd, z, wd, order)
        dx_w, dw_w, _ = H.backward(gd, z, wd, order, H.FLAG_WHOLE_IMAGE)
        assert torch.equal(dx, dx_w), (Hh, C, K, order)
        u_o = oracle.dy(g.astype(np.float64), w.astype(np.float64), 0, order, nthreads=8)
        assert rel_err(host(dx), u_o) < TOL, (Hh, C, K, order)
