"""GPU: ActNorm / Squeeze / Coupling of libinvflow_hip (csrc/glow_step.hip) through the C ABI
against the reference's golden vectors and the oracle, and the host layers built on them against the reference
layers' recorded outputs and gradients."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def H():
    import invflow_hip
    invflow_hip.lib()
    return invflow_hip


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("path", golden_files("actnorm_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_actnorm(H, path):
    g = load_golden(path)
    x, t, ls = dev(g["x"]), dev(g["translation"]), dev(g["log_scale"])
    if int(g["init_from_data"]):
        mean, lstd = H.actnorm_stats(x)
        assert rel_err(host(mean), g["translation"]) < TOL and rel_err(host(lstd), g["log_scale"]) < TOL
    y, ld = H.actnorm(x, t, ls)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < TOL
    assert rel_err(host(H.actnorm(dev(g["y"]), t, ls, reverse=True)), g["x_rev"]) < TOL
    gx, gt, gls = H.actnorm_backward(dev(g["gy"]), dev(g["gld"]), x, t, ls)
    assert rel_err(host(gx), g["gx"]) < TOL and rel_err(host(gt), g["g_translation"]) < TOL
    assert rel_err(host(gls), g["g_log_scale"]) < TOL


@pytest.mark.parametrize("path", golden_files("squeeze_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_squeeze(H, path):
    g = load_golden(path)
    y = H.space_to_depth(dev(g["x"]))
    assert np.array_equal(host(y), g["y"])
    assert np.array_equal(host(H.depth_to_space(y)), g["x"])


@pytest.mark.parametrize("path", golden_files("coupling_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_coupling(H, path):
    g = load_golden(path)
    x, h = dev(g["x"]), dev(g["h"])
    y, ld = H.coupling(x, h)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < TOL
    assert rel_err(host(H.coupling(dev(g["y"]), h, reverse=True)), g["x_rev"]) < TOL
    gx, gh = H.coupling_backward(dev(g["gy"]), dev(g["gld"]), x, h)
    ch = x.shape[1] // 2
    assert rel_err(host(gh), g["gh"]) < TOL
    assert rel_err(host(gx)[:, ch:], g["gx_total"][:, ch:]) < TOL and np.array_equal(host(gx)[:, :ch], g["gy"][:, :ch])


SHAPES = [(5, 7, 7, 7), (3, 4, 9, 6), (16, 64, 32, 32), (2, 256, 8, 8), (1, 2, 2, 2), (7, 10, 28, 28)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_against_oracle(H, oracle, shape):
    """ragged planes (H*W not a multiple of 4: the scalar path), one-pixel-row images, the north-star plane size and a
    wide layer: every op against the fp64 restatement; in-place forms; empty batch."""
    B, C, Hh, Ww = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32) * 2 + 0.5
    t = rng.standard_normal(C).astype(np.float32)
    ls = (rng.standard_normal(C) * 0.7).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    gld = rng.standard_normal(B).astype(np.float32)
    y, ld = H.actnorm(dev(x), dev(t), dev(ls))
    y_o, ld_o = oracle.actnorm_forward(x, t, ls)
    assert rel_err(host(y), y_o) < TOL and rel_err(host(ld), ld_o) < TOL
    assert rel_err(host(H.actnorm(y, dev(t), dev(ls), reverse=True)), x) < TOL  # round trip
    gx, gt, gls = H.actnorm_backward(dev(gy), dev(gld), dev(x), dev(t), dev(ls))
    gx_o, gt_o, gls_o = oracle.actnorm_backward(gy, gld, x, t, ls)
    assert rel_err(host(gx), gx_o) < TOL and rel_err(host(gt), gt_o) < TOL and rel_err(host(gls), gls_o) < TOL
    if B * Hh * Ww > 1:
        m, s = H.actnorm_stats(dev(x))
        m_o, s_o = oracle.actnorm_stats(x)
        assert rel_err(host(m), m_o) < TOL and rel_err(host(s), s_o) < TOL
    if Hh % 2 == 0 and Ww % 2 == 0:
        q = H.space_to_depth(dev(x))
        assert np.array_equal(host(q), oracle.space_to_depth(x))
        assert np.array_equal(host(H.depth_to_space(q)), x)
    if C % 2 == 0:
        h = rng.standard_normal(shape).astype(np.float32) * 1.5
        z, zl = H.coupling(dev(x), dev(h))
        z_o, zl_o = oracle.coupling_forward(x, h)
        assert rel_err(host(z), z_o) < TOL and rel_err(host(zl), zl_o) < TOL
        assert rel_err(host(H.coupling(z, dev(h), reverse=True)), x) < TOL
        gx, gh = H.coupling_backward(dev(gy), dev(gld), dev(x), dev(h))
        gx_o, gh_o = oracle.coupling_backward(gy, gld, x, h)
        assert rel_err(host(gx), gx_o) < TOL and rel_err(host(gh), gh_o) < TOL


def test_errors_and_empty(H):
    e = torch.empty(0, 4, 6, 6, device="cuda")
    p = torch.zeros(4, device="cuda")
    y, ld = H.actnorm(e, p, p)
    assert y.shape == e.shape and ld.shape == (0,)
    with pytest.raises(RuntimeError, match="must be even"):
        H.space_to_depth(torch.zeros(1, 2, 5, 6, device="cuda"))
    with pytest.raises(RuntimeError, match="must be even"):
        H.coupling(torch.zeros(1, 3, 4, 4, device="cuda"), torch.zeros(1, 3, 4, 4, device="cuda"))
    with pytest.raises(RuntimeError):
        H.actnorm(torch.zeros(1, 3, 4, 4, device="cuda"), torch.zeros(2, device="cuda"), torch.zeros(3, device="cuda"))


# ---- host layers (inf/layers/actnorm.py, squeeze.py, coupling.py of this package) ----------------------------------
def test_actnorm_layer(H):
    from inf.layers.actnorm import ActNorm
    g = load_golden(golden_files("actnorm_b4c5_7x5_datainit")[0])
    layer = ActNorm(5).cuda()
    x = dev(g["x"]).requires_grad_(True)
    y, ld = layer(x)  # first call: data-dependent initialisation
    assert int(layer.initialized) == 1
    assert rel_err(host(layer.translation), g["translation"]) < TOL and rel_err(host(layer.log_scale), g["log_scale"]) < TOL
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < TOL
    ((y * dev(g["gy"])).sum() + (ld * dev(g["gld"])).sum()).backward()
    assert rel_err(host(x.grad), g["gx"]) < TOL
    assert rel_err(host(layer.translation.grad), g["g_translation"]) < TOL
    assert rel_err(host(layer.log_scale.grad), g["g_log_scale"]) < TOL
    with torch.no_grad():
        assert rel_err(host(layer.reverse(y.detach())), g["x_rev"]) < TOL
    assert rel_err(host(layer.logdet(x)), g["logdet"]) < TOL


def test_squeeze_layer(H):
    from inf.layers.squeeze import Squeeze, UnSqueeze
    g = load_golden(golden_files("squeeze_b2c3_8x12")[0])
    x = dev(g["x"]).requires_grad_(True)
    y, ld = Squeeze()(x)
    assert np.array_equal(host(y), g["y"]) and float(ld.abs().sum()) == 0.0 and ld.shape == (2,)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert torch.equal(x.grad, UnSqueeze()(w)[0])  # the gradient of a permutation is the inverse permutation
    assert np.array_equal(host(Squeeze().reverse(y.detach())), g["x"])


@pytest.mark.parametrize("path", golden_files("coupling_"), ids=lambda p: p.split("/")[-1][:-4])
def test_coupling_layer(H, path):
    """the reference layer's state_dict loads as is; outputs, the FULL input gradient (direct part + the part through
    the conditioner) and the conditioner's parameter gradients follow"""
    from inf.layers.coupling import Coupling
    g = load_golden(path)
    B, C, Hh, Ww = g["x"].shape
    layer = Coupling((C, Hh, Ww), width=int(g["width"]))
    sd = {k.replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("net__")}
    layer.load_state_dict(sd)
    layer = layer.cuda()
    x = dev(g["x"]).requires_grad_(True)
    y, ld = layer(x)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < TOL
    ((y * dev(g["gy"])).sum() + (ld * dev(g["gld"])).sum()).backward()
    assert rel_err(host(x.grad), g["gx_total"]) < 2e-5
    with torch.no_grad():
        assert rel_err(host(layer.reverse(y.detach())), g["x_rev"]) < TOL
    # torch-expression path of the same layer (CPU tensors) gives the same parameter gradients
    ref = Coupling((C, Hh, Ww), width=int(g["width"]))
    ref.load_state_dict(sd)
    xc = torch.from_numpy(g["x"]).requires_grad_(True)
    yc, lc = ref(xc)
    ((yc * torch.from_numpy(g["gy"])).sum() + (lc * torch.from_numpy(g["gld"])).sum()).backward()
    for (k, p), (_, q) in zip(layer.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            assert rel_err(host(p.grad), q.grad.numpy()) < 1e-4, k


def test_full_size_round_trips(H):
    """north-star activation size (128, 64, 32, 32): forward then reverse of each op is the identity to fp32 rounding;
    log-dets add up as the definitions say"""
    torch.manual_seed(0)
    x = torch.randn(128, 64, 32, 32, device="cuda")
    t, ls = torch.randn(64, device="cuda"), torch.randn(64, device="cuda") * 0.3
    y, ld = H.actnorm(x, t, ls)
    assert float((H.actnorm(y, t, ls, reverse=True) - x).abs().max()) < 1e-5
    assert abs(float(ld[0]) + 1024 * float(ls.double().sum())) < 1e-2 and float(ld.max() - ld.min()) == 0.0
    q = H.space_to_depth(x)
    assert q.shape == (128, 256, 16, 16) and torch.equal(H.depth_to_space(q), x)
    h = torch.randn_like(x)
    z, zl = H.coupling(x, h)
    assert float((H.coupling(z, h, reverse=True) - x).abs().max()) < 2e-5
    ref = (2 * torch.tanh(h[:, ::2].double() / 2)).flatten(1).sum(-1)
    assert float(((zl.double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 1e-5


# ---- activations (SmoothLeakyRelu, SplineActivation with shared weights) ---------------------------------------------
@pytest.mark.parametrize("path", golden_files("slr_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_slr(H, path):
    from inf.layers.activations import SmoothLeakyRelu
    g = load_golden(path)
    a = float(g["alpha"])
    y, ld = H.slr(dev(g["x"]), a)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < TOL
    assert rel_err(host(H.slr(dev(g["y"]), a, reverse=True)), g["x_rev"]) < TOL
    assert rel_err(host(H.slr_backward(dev(g["gy"]), dev(g["gld"]), dev(g["x"]), a)), g["gx"]) < TOL
    layer = SmoothLeakyRelu(a)
    x = dev(g["x"]).requires_grad_(True)
    yl, ll = layer(x)
    ((yl * dev(g["gy"])).sum() + (ll * dev(g["gld"])).sum()).backward()
    assert rel_err(host(yl), g["y"]) < TOL and rel_err(host(x.grad), g["gx"]) < TOL
    with torch.no_grad():
        assert rel_err(host(layer.reverse(yl.detach())), g["x_rev"]) < TOL


@pytest.mark.parametrize("path", golden_files("spline_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_spline(H, oracle, path):
    """the reference layer's parameters loaded as they are: outputs, log-det, reverse, the input gradient and the
    gradients of the three parameter vectors (through the knot tables) against the reference's autograd"""
    from inf.layers.activations import SplineActivation
    g = load_golden(path)
    nb, tb = int(g["n_bins"]), float(g["tail_bound"])
    layer = SplineActivation(g["x"].shape[1:], n_bins=nb, tail_bound=tb)
    layer.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p_")})
    layer = layer.cuda()
    x = dev(g["x"]).requires_grad_(True)
    y, ld = layer(x)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < 2e-5
    ((y * dev(g["gy"])).sum() + (ld * dev(g["gld"])).sum()).backward()
    assert rel_err(host(x.grad), g["gx"]) < 2e-5
    for k, p in layer.named_parameters():
        assert rel_err(host(p.grad), g["g_" + k]) < 1e-4, k
    with torch.no_grad():
        xr = layer.reverse(y.detach())
    assert rel_err(host(xr), g["x_rev"]) < TOL and rel_err(host(xr), g["x"]) < TOL
    # the C-ABI ops against the oracle's tables and values
    cw, ch, dv = oracle.spline_tables(g["p_unnormalized_widths"], g["p_unnormalized_heights"], g["p_unnormalized_derivatives"], tb)
    y2, ld2 = H.rqspline(dev(g["x"]), dev(cw), dev(ch), dev(dv), tb)
    y_o, lad_o = oracle.rqspline(g["x"], cw, ch, dv, tb)
    assert rel_err(host(y2), y_o) < TOL and rel_err(host(ld2), lad_o.reshape(len(y_o), -1).sum(-1)) < 2e-5


@pytest.mark.parametrize("path", golden_files("splinepe_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_spline_individual_weights(H, oracle, path):
    """SplineActivation(individual_weights=True): the reference layer's per-element parameters loaded as they are; outputs,
    log-det, reverse, the input gradient and the gradients of the three parameter tensors (summed over the batch) against
    the reference's autograd; the C-ABI ops against the oracle; the same layer under bf16 autocast computes in fp32"""
    from inf.layers.activations import SplineActivation
    g = load_golden(path)
    nb, tb = int(g["n_bins"]), float(g["tail_bound"])
    layer = SplineActivation(g["x"].shape[1:], n_bins=nb, tail_bound=tb, individual_weights=True)
    layer.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p_")})
    layer = layer.cuda()
    assert layer._hip_pe(dev(g["x"]))
    x = dev(g["x"]).requires_grad_(True)
    y, ld = layer(x)
    assert rel_err(host(y), g["y"]) < TOL and rel_err(host(ld), g["logdet"]) < 2e-5
    ((y * dev(g["gy"])).sum() + (ld * dev(g["gld"])).sum()).backward()
    assert rel_err(host(x.grad), g["gx"]) < 2e-5
    for k, p in layer.named_parameters():
        assert p.grad.shape == p.shape and rel_err(host(p.grad), g["g_" + k]) < 1e-4, k
    with torch.no_grad():
        xr = layer.reverse(y.detach())
    assert rel_err(host(xr), g["x_rev"]) < TOL and rel_err(host(xr), g["x"]) < TOL
    uw, uh, ud = (dev(g["p_unnormalized_" + n]) for n in ("widths", "heights", "derivatives"))
    y2, ld2 = H.rqspline_pe(dev(g["x"]), uw, uh, ud, tb)
    y_o, lad_o = oracle.rqspline_individual(g["x"], host(uw), host(uh), host(ud), tb)
    assert rel_err(host(y2), y_o) < TOL and rel_err(host(ld2), lad_o.reshape(len(y_o), -1).sum(-1)) < 2e-5
    xi, ldi = H.rqspline_pe(y2, uw, uh, ud, tb, inverse=True)
    assert rel_err(host(xi), g["x"]) < TOL and rel_err(host(ldi), -host(ld2)) < 1e-4
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y3, _ = layer(dev(g["x"]))
    assert y3.dtype == torch.float32 and torch.equal(y3, y.detach())
    # empty batch
    ye, lde = H.rqspline_pe(dev(g["x"])[:0], uw, uh, ud, tb)
    assert ye.shape[0] == 0 and lde.shape[0] == 0


@pytest.mark.parametrize("shape", [(4, 6, 7, 5), (16, 64, 32, 32), (3, 12, 16, 16)], ids=lambda s: "x".join(map(str, s)))
def test_activations_against_oracle(H, oracle, shape):
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape) * 4).astype(np.float32)
    gy = rng.standard_normal(shape).astype(np.float32)
    gld = rng.standard_normal(shape[0]).astype(np.float32)
    y, ld = H.slr(dev(x), 0.3)
    y_o, ld_o = oracle.slr_forward(x, 0.3)
    assert rel_err(host(y), y_o) < TOL and rel_err(host(ld), ld_o) < TOL
    assert rel_err(host(H.slr(y, 0.3, reverse=True)), x) < TOL
    assert rel_err(host(H.slr_backward(dev(gy), dev(gld), dev(x), 0.3)), oracle.slr_backward(gy, gld, x, 0.3)) < TOL
    uw, uh, ud = rng.standard_normal(5), rng.standard_normal(5), rng.standard_normal(4)
    cw, ch, dv = oracle.spline_tables(uw, uh, ud, 6.0)
    s, sl = H.rqspline(dev(x), dev(cw), dev(ch), dev(dv), 6.0)
    s_o, lad_o = oracle.rqspline(x, cw, ch, dv, 6.0)
    assert rel_err(host(s), s_o) < TOL and rel_err(host(sl), lad_o.reshape(shape[0], -1).sum(-1)) < 2e-5
    back, _ = H.rqspline(s, dev(cw), dev(ch), dev(dv), 6.0, inverse=True)
    assert rel_err(host(back), x) < 2e-5
    # gradients against autograd through this package's torch expressions of the same spline in float64 on the CPU
    from inf.layers.activations import SplineActivation, _spline_torch
    ref = SplineActivation(shape[1:], n_bins=5, tail_bound=6.0).double()
    with torch.no_grad():
        ref.unnormalized_widths.copy_(torch.from_numpy(uw)); ref.unnormalized_heights.copy_(torch.from_numpy(uh))
        ref.unnormalized_derivatives.copy_(torch.from_numpy(ud))
    xc = torch.from_numpy(x).double().requires_grad_(True)
    yc, lc = _spline_torch(ref, xc, inverse=False)
    ((yc * torch.from_numpy(gy).double()).sum() + (lc * torch.from_numpy(gld).double()).sum()).backward()
    layer = SplineActivation(shape[1:], n_bins=5, tail_bound=6.0)
    layer.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    layer = layer.cuda()
    xg = dev(x).requires_grad_(True)
    yg, lg = layer(xg)
    ((yg * dev(gy)).sum() + (lg * dev(gld)).sum()).backward()
    # (fp32 derivative arithmetic against a float64 truth: the log-derivative terms divide by small differences)
    assert rel_err(host(xg.grad), xc.grad.numpy()) < 5e-5
    for (k, p), (_, q) in zip(layer.named_parameters(), ref.named_parameters()):
        assert rel_err(host(p.grad), q.grad.numpy()) < 1e-4, k


@pytest.mark.parametrize("nb,tb", [(10, 20.0), (16, 8.0), (9, 3.0)], ids=str)
def test_shared_spline_of_the_32x32_builders(H, oracle, nb, tb):
    """The 32x32x3 models' activation: ONE set of knots, 10 bins, tail bound 20 (if_glow_cifar.py:23-26) -- more bins than the
    per-element spline of the MNIST model.  Library kernels against the oracle (values, log-derivative, inverse) and against
    autograd through the torch expressions of the same spline in float64 (input and parameter gradients)."""
    from inf.layers.activations import SplineActivation, _spline_torch
    shape = (3, 12, 16, 16)
    rng = np.random.default_rng(nb)
    x = (rng.standard_normal(shape) * 0.6 * tb).astype(np.float32)  # (some elements in the linear tails)
    gy = rng.standard_normal(shape).astype(np.float32)
    gld = rng.standard_normal(shape[0]).astype(np.float32)
    uw, uh, ud = rng.standard_normal(nb), rng.standard_normal(nb), rng.standard_normal(nb - 1)
    cw, ch, dv = oracle.spline_tables(uw, uh, ud, tb)
    s, sl = H.rqspline(dev(x), dev(cw), dev(ch), dev(dv), tb)
    s_o, lad_o = oracle.rqspline(x, cw, ch, dv, tb)
    assert rel_err(host(s), s_o) < TOL and rel_err(host(sl), lad_o.reshape(shape[0], -1).sum(-1)) < 2e-5
    back, _ = H.rqspline(s, dev(cw), dev(ch), dev(dv), tb, inverse=True)
    assert rel_err(host(back), x) < 2e-5
    ref = SplineActivation(shape[1:], n_bins=nb, tail_bound=tb).double()
    with torch.no_grad():
        ref.unnormalized_widths.copy_(torch.from_numpy(uw)); ref.unnormalized_heights.copy_(torch.from_numpy(uh))
        ref.unnormalized_derivatives.copy_(torch.from_numpy(ud))
    xc = torch.from_numpy(x).double().requires_grad_(True)
    yc, lc = _spline_torch(ref, xc, inverse=False)
    ((yc * torch.from_numpy(gy).double()).sum() + (lc * torch.from_numpy(gld).double()).sum()).backward()
    layer = SplineActivation(shape[1:], n_bins=nb, tail_bound=tb)
    layer.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    layer = layer.cuda()
    assert layer._hip(dev(x))  # the library path, not the torch expressions
    xg = dev(x).requires_grad_(True)
    yg, lg = layer(xg)
    assert rel_err(host(yg), yc.detach().numpy()) < TOL
    ((yg * dev(gy)).sum() + (lg * dev(gld)).sum()).backward()
    # (fp32 derivative arithmetic against a float64 truth: the log-derivative terms divide by bin widths, and sixteen random
    # bins on [-8, 8] include some of a few hundredths)
    assert rel_err(host(xg.grad), xc.grad.numpy()) < (5e-5 if nb <= 10 else 5e-4)
    for (k, p), (_, q) in zip(layer.named_parameters(), ref.named_parameters()):
        assert rel_err(host(p.grad), q.grad.numpy()) < (2e-4 if nb <= 10 else 2e-3), k


def test_spline_tables_kernels(H, oracle):
    """knot tables and their way back on the library against the oracle's tables and autograd through this package's
    torch expressions of the same formulas"""
    from inf.layers.activations import spline_tables
    rng = np.random.default_rng(5)
    for nb, tb in [(5, 10.0), (8, 3.0), (2, 1.5), (10, 20.0), (16, 4.0)]:
        uw, uh, ud = rng.standard_normal(nb), rng.standard_normal(nb), rng.standard_normal(nb - 1)
        cw, ch, dv = H.rqspline_tables(dev(uw), dev(uh), dev(ud), tb)
        cw_o, ch_o, dv_o = oracle.spline_tables(uw, uh, ud, tb)
        assert rel_err(host(cw), cw_o) < TOL and rel_err(host(ch), ch_o) < TOL and rel_err(host(dv), dv_o) < TOL
        tw, th, td = (torch.from_numpy(a).double().requires_grad_(True) for a in (uw, uh, ud))
        tcw, tch, tdv = spline_tables(tw, th, td, tb)
        gt = rng.standard_normal((3, nb + 1))
        ((tcw * torch.from_numpy(gt[0])).sum() + (tch * torch.from_numpy(gt[1])).sum() + (tdv * torch.from_numpy(gt[2])).sum()).backward()
        guw, guh, gud = H.rqspline_tables_backward(dev(gt), dev(uw), dev(uh), dev(ud), tb)
        assert rel_err(host(guw), tw.grad.numpy()) < TOL and rel_err(host(guh), th.grad.numpy()) < TOL
        assert rel_err(host(gud), td.grad.numpy()) < TOL


@pytest.mark.parametrize("nb,tb,shape", [(10, 20.0, (3, 12, 16, 16)), (5, 10.0, (2, 24, 8, 8)), (16, 3.0, (1, 4, 5, 7)), (2, 1.0, (4, 2, 3, 3)),
                                         (7, 6.0, (2, 40, 32, 32))])
def test_spline_from_parameters_is_the_two_call_form_bit_for_bit(H, nb, tb, shape):
    """ifl_rqspline_p_f32 / _backward_f32 (knot tables computed inside the spline's launch, their gradients chained to the
    parameters inside the reduction's launch) against tables + spline as two calls each way: the same arithmetic in the same
    order -- values, log-det, inverse, input gradient and the three parameter gradients identical in every bit."""
    from inf.layers.activations import _SplineFn, _SplineParamFn, _TablesFn
    torch.manual_seed(nb)
    x = (torch.randn(shape) * 0.6 * tb).cuda()
    gy, gld = torch.randn(shape).cuda(), torch.randn(shape[0]).cuda()
    ps = [torch.randn(nb).cuda().requires_grad_(), torch.randn(nb).cuda().requires_grad_(), torch.randn(nb - 1).cuda().requires_grad_()]
    outs = []
    for merged in (True, False):
        for p in ps:
            p.grad = None
        xg = x.clone().requires_grad_()
        if merged:
            y, ld = _SplineParamFn.apply(xg, *ps, tb)
        else:
            y, ld = _SplineFn.apply(xg, *_TablesFn.apply(*ps, tb), tb)
        ((y * gy).sum() + (ld * gld).sum()).backward()
        outs.append([y.detach(), ld.detach(), xg.grad] + [p.grad.clone() for p in ps])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    back = H.rqspline_p(outs[0][0].contiguous(), *[p.detach() for p in ps], tb, inverse=True, want_logdet=False)[0]
    cw, ch, dv = H.rqspline_tables(*[p.detach() for p in ps], tb)
    assert torch.equal(back, H.rqspline(outs[0][0].contiguous(), cw, ch, dv, tb, inverse=True, want_logdet=False)[0])

