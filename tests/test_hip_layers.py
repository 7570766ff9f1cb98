"""GPU: the `inf` operator surface on top of the HIP library -- layers, autograd, FlowSequential,
the inv_conv_with_bp drop-in module, SelfNormConv -- against the oracle / golden vectors."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def H():
    assert torch.cuda.is_available()
    import invflow_hip
    invflow_hip.lib()
    return invflow_hip


def host(t):
    return t.detach().cpu().double().numpy()


def check_inverse(module, data_dim):
    """tests/inf/test_layers.py:19-36 (check_inverse): forward then reverse, atol 1e-3."""
    module.reset_parameters()
    module.to("cuda")
    x = torch.randn(data_dim).cuda()
    fwd, logdet = module(x)
    rev = module.reverse(fwd)
    np.testing.assert_allclose(x.cpu().numpy(), rev.detach().cpu().view(data_dim).numpy(), atol=1e-3)
    return logdet


def test_reference_test_inv_conv(H):
    """tests/inf/test_layers.py:182-190 (test_inv_conv): (1,4,5,5), 3x3, no_pad and with_pad TL."""
    from inf.layers.inv_conv import inv_flow_no_pad, inv_flow_with_pad
    torch.manual_seed(0)
    size = (1, 4, 5, 5)
    assert check_inverse(inv_flow_no_pad(4, 4, (3, 3)), size) == 0.0
    assert check_inverse(inv_flow_with_pad(4, 4, (3, 3), order="TL"), size) == 0.0
    check_inverse(inv_flow_with_pad(4, 4, (2, 2), order="TL"), size)  # test_layers.py:152
    for order in ("TR", "BL", "BR"):
        check_inverse(inv_flow_with_pad(4, 4, (3, 3), order=order), size)
    for C, hw in ((64, 32), (32, 16)):  # MFMA path
        for order in ("TL", "TR", "BL", "BR"):
            check_inverse(inv_flow_with_pad(C, C, (3, 3), order=order), (3, C, hw, hw))


@pytest.mark.parametrize("order", ["TL", "TR", "BL", "BR"])
@pytest.mark.parametrize("shape", [(2, 6, 7, 5, 3), (3, 64, 32, 32, 3), (2, 32, 12, 16, 2)])
def test_layer_autograd_matches_oracle(H, oracle, order, shape):
    from inf.layers.inv_conv import inv_flow_with_pad
    B, C, Hh, Ww, K = shape
    torch.manual_seed(hash((order, shape)) % 1000)
    layer = inv_flow_with_pad(C, C, (K, K), order=order).cuda()
    with torch.no_grad():
        layer.weight_fwd.add_(0.02 * torch.randn_like(layer.weight_fwd))
    x = torch.randn(B, C, Hh, Ww, device="cuda", requires_grad=True)
    g = torch.randn(B, C, Hh, Ww, device="cuda")
    z, ldj = layer(x)
    assert ldj == 0.0
    (z * g).sum().backward()
    w64, x64, g64 = host(layer.weight_fwd), host(x), host(g)
    z_o = oracle.inverse(x64, w64, 0, order, nthreads=8)
    u_o = oracle.dy(g64, w64, 0, order, nthreads=8)
    dw_o = oracle.dw(z_o, u_o, (K, K), 0, order, nthreads=8)
    assert rel_err(host(z), z_o) < TOL
    assert rel_err(host(x.grad), u_o) < TOL
    # the boosted weights make A badly conditioned (|z| up to 1e4..1e6 at C=64): dW is a difference of large
    # numbers there, so its bound is looser than the 1e-5 of z / dx at the reference init
    assert rel_err(host(layer.weight_fwd.grad), dw_o) < 3 * TOL
    # the gradient already carries the mask of reset_gradients (inv_conv.py:223-230)
    before = layer.weight_fwd.grad.clone()
    layer.reset_gradients()
    assert torch.equal(before, layer.weight_fwd.grad)
    # the MFMA kernels and the general kernels are two statements of the same op
    if C % 32 == 0:
        z2 = H.inverse(x.detach(), layer.weight_fwd.detach(), order, H.FLAG_NO_MFMA)
        dx2, dw2, _ = H.backward(g, z2, layer.weight_fwd.detach(), order, H.FLAG_NO_MFMA)
        assert rel_err(host(z2), z_o) < TOL and rel_err(host(dx2), u_o) < TOL and rel_err(host(dw2), dw_o) < 3 * TOL


# BASELINE.json configs as (batch, C, H, W, K) of their inverse-flow layers: if_glow_mnist (configs[2]: batch 100, 2x2 kernels
# after the squeezes, inf/experiments/if_glow_mnist.py:156-159,190), if_glow_cifar per GPU of 8 (configs[3]: 256/8 = 32,
# C = 12/24/48, inf/experiments/if_glow_cifar.py:48-50,94-96), if_multiGPU_imagenet32 (configs[4]: C = 256)
CONFIG_LAYERS = [(100, 4, 14, 14, 2), (100, 8, 7, 7, 2), (32, 12, 16, 16, 3), (32, 24, 8, 8, 3), (32, 48, 4, 4, 3), (16, 256, 8, 8, 3)]


@pytest.mark.parametrize("shape", CONFIG_LAYERS, ids=lambda s: "b%dc%d_%dx%d_k%d" % s)
def test_config_layer_shapes_through_flowsequential(H, oracle, shape):
    """The layers of BASELINE.json's model configs at the configs' own batch sizes, through the layer's autograd and a
    FlowSequential (log-det counted once, log p = base log-prob), against the fp64 oracle."""
    from inf.layers.flowsequential import FlowSequential
    from inf.layers.inv_conv import inv_flow_no_pad
    from inf.train.losses import NegativeGaussianLoss
    B, C, Hh, Ww, K = shape
    torch.manual_seed(B + C)
    layer = inv_flow_no_pad(C, C, (K, K)).cuda()
    model = FlowSequential(NegativeGaussianLoss((C, Hh, Ww)), layer).cuda()
    x = torch.randn(B, C, Hh, Ww, device="cuda", requires_grad=True)
    z, logp = model(x)
    assert torch.allclose(logp, model.base_distribution.log_prob(z))
    (-logp.mean()).backward()
    w64, x64 = host(layer.weight_fwd), host(x)
    z_o = oracle.inverse(x64, w64, 0, "TL", nthreads=8)
    assert rel_err(host(z), z_o) < TOL
    # d(-mean log p)/dz = z / B for the standard normal base
    g_o = z_o / B
    u_o = oracle.dy(g_o, w64, 0, "TL", nthreads=8)
    dw_o = oracle.dw(z_o, u_o, (K, K), 0, "TL", nthreads=8)
    assert rel_err(host(x.grad), u_o) < TOL
    assert rel_err(host(layer.weight_fwd.grad), dw_o) < 3 * TOL
    assert rel_err(host(model.reconstruct(x.detach())), x64) < 1e-4


def test_inv_flow_unit_and_sequential(H):
    from inf.layers.flowsequential import FlowSequential
    from inf.layers.inv_conv import inv_flow_no_pad
    from inf.layers.inv_flow import Inv_FlowUnit
    from inf.train.losses import NegativeGaussianLoss
    torch.manual_seed(3)
    C, S = 8, 8
    unit = Inv_FlowUnit(C, C, (3, 3)).cuda()
    x = torch.randn(5, C, S, S, device="cuda")
    y, ld = unit(x)
    assert ld == 0.0
    np.testing.assert_allclose(host(unit.reverse(y)), host(x), atol=1e-3)
    model = FlowSequential(NegativeGaussianLoss((C, S, S)), inv_flow_no_pad(C, C, (3, 3)), unit,
                           inv_flow_no_pad(C, C, (2, 2))).cuda()
    out, lp = model(x)
    assert torch.allclose(lp, model.base_distribution.log_prob(out))  # every layer has log-det 0, counted once
    assert rel_err(host(model.reconstruct(x)), host(x)) < 1e-3  # six inverse + six forward layers in fp32
    samples, _ = model.sample(4)
    assert samples.shape == (4, C, S, S) and torch.isfinite(samples).all()
    loss = -lp.mean()
    loss.backward()
    ps = list(model.parameters())
    assert len(ps) == 6 and all(p.grad is not None and torch.isfinite(p.grad).all() for p in ps)


def test_drop_in_extension_module(H, oracle):
    """The four functions of the reference pybind module (inv_conv_with_bp_general.cpp:115-120) with the
    reference calling convention: caller-allocated outputs, list return, scratch M ignored."""
    import inv_conv_with_bp as ext
    rng = np.random.default_rng(11)
    B, C, S, K = 2, 8, 6, 3
    x = torch.tensor(rng.standard_normal((B, C, S, S)), dtype=torch.float32).cuda()
    g = torch.tensor(rng.standard_normal((B, C, S, S)), dtype=torch.float32).cuda()
    w = torch.tensor(rng.standard_normal((C, C, K, K)) * 0.05, dtype=torch.float32).cuda()
    out = x * 0.0  # inv_conv.py:48
    z = ext.inverse(x, w, out)
    assert isinstance(z, list) and z[0] is out
    z_o = oracle.inverse(host(x), host(w))
    assert rel_err(host(out), z_o) < TOL
    rev = ext.forward(out, w, torch.zeros_like(x))
    assert rel_err(host(rev[0]), host(x)) < TOL
    M_dy = torch.zeros_like(x)
    M_dk = torch.zeros((B, C, K, K, S, S), device="cuda")  # inv_conv.py:70
    dyv = ext.dy(g, w, M_dy, torch.zeros_like(x))
    u_o = oracle.dy(host(g), host(w))
    assert rel_err(host(dyv[0]), u_o) < TOL
    dk = ext.dw(x, w, g, M_dk, torch.zeros_like(w))
    assert rel_err(host(dk[0]), oracle.dw(z_o, u_o, (K, K))) < TOL
    with pytest.raises(RuntimeError, match="must be contiguous"):
        ext.inverse(x.transpose(2, 3), w, out)


@pytest.mark.parametrize("path", golden_files("selfnorm_"), ids=lambda p: p.split("/")[-1][:-4])
def test_selfnorm_matches_golden(H, path):
    from inf.layers.selfnorm import selfnorm_conv_2d
    g = load_golden(path)
    p = (g["pad"], g["pad"])
    t = lambda a: torch.tensor(a, dtype=torch.float32).cuda()  # noqa: E731
    x = t(g["x"]).requires_grad_(True)
    W = t(g["w"]).requires_grad_(True)
    R = t(g["r"]).requires_grad_(True)
    b = t(g["bias"]).requires_grad_(True) if "bias" in g else None
    z = selfnorm_conv_2d(x, W, b, R, (1, 1), p)
    assert rel_err(host(z), g["z"]) < TOL
    z.backward(t(g["gz"]))
    assert rel_err(host(x.grad), g["dx"]) < TOL
    assert rel_err(host(W.grad), g["dw_fwd"]) < TOL
    assert rel_err(host(R.grad), g["dw_inv"]) < TOL
    if b is not None:
        assert rel_err(host(b.grad), g["dbias"]) < TOL


@pytest.mark.parametrize("sym", [False, True])
def test_selfnorm_layer_recon_and_exact(H, sym):
    from inf.layers.selfnorm import SelfNormConv, SelfNormFC
    g = load_golden(golden_files("selfnorm_b3c4_8x8_k3_p1")[0])
    t = lambda a: torch.tensor(a, dtype=torch.float32).cuda()  # noqa: E731
    layer = SelfNormConv(4, 4, (3, 3), bias=False, padding=1, sym_recon_grad=sym).cuda()
    with torch.no_grad():
        layer.weight_fwd.copy_(t(g["w"]))
        layer.weight_inv.copy_(t(g["r"]))
    x = t(g["x"])
    layer(x)
    loss = layer.add_recon_grad()
    tag = "sym" if sym else "asym"
    assert abs(float(loss) - g["recon_loss_" + tag]) < 1e-4 * abs(g["recon_loss_" + tag])
    assert rel_err(host(layer.weight_fwd.grad), g["recon_dw_" + tag]) < 1e-4
    assert rel_err(host(layer.weight_inv.grad), g["recon_dr_" + tag]) < 1e-4
    # exact mode (tests/inf/test_layers.py:39-46, 101-108): dense inverse round trip and slogdet
    out, ld = layer(x, compute_expensive=True)
    rev = layer.reverse(out, compute_expensive=True)
    np.testing.assert_allclose(host(rev), host(x), atol=1e-3)
    J = torch.autograd.functional.jacobian(lambda a: torch.nn.functional.conv2d(a[None], layer.weight_fwd.detach().cpu().double(), None, 1, 1)[0],
                                           x[0].cpu().double())
    ld_ref = torch.slogdet(J.reshape(256, 256))[1]
    np.testing.assert_allclose(host(ld)[0], float(ld_ref), atol=1e-4)
    # the exact log-det is part of the loss of the exact-gradient baseline (experiment.py:161): its gradient reaches
    # weight_fwd and equals that of slogdet of the dense operator (selfnorm.py:240-246)
    layer.weight_fwd.grad = None
    layer.logdet(x, compute_expensive=True)[0].backward()
    w64 = layer.weight_fwd.detach().cpu().double().requires_grad_(True)
    eye = torch.eye(256, dtype=torch.float64).view(256, 4, 8, 8)
    torch.slogdet(torch.nn.functional.conv2d(eye, w64, None, 1, 1).flatten(start_dim=1).T)[1].backward()
    assert rel_err(host(layer.weight_fwd.grad), w64.grad.numpy()) < 1e-4
    fc = SelfNormFC(12, 12).cuda()
    xf = torch.randn(5, 12, device="cuda")
    of, ldf = fc(xf, compute_expensive=True)
    np.testing.assert_allclose(host(fc.reverse(of, compute_expensive=True)), host(xf), atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 32, 32, 3), (2, 32, 16, 16, 3), (2, 64, 16, 16, 2), (2, 6, 9, 5, 3)],
                         ids=lambda s_: "b%dc%d_%dx%d_k%d" % s_)
def test_flow_unit_fused_block(H, shape):
    """Inv_FlowUnit (inf/layers/inv_flow.py:13-53) as one library call (ifl_unit_inverse_f32 / ifl_unit_backward_f32):
    same outputs and gradients as the four layers called one after the other (bitwise on the MFMA path: the same
    kernels run on the same data), and the oracle's chain within the tolerance."""
    from inf.layers.inv_flow import Inv_FlowUnit
    from oracle import oracle as O
    B, C, Hh, Ww, K = shape
    torch.manual_seed(3)
    unit = Inv_FlowUnit(C, C, (K, K)).cuda()
    x = torch.randn(B, C, Hh, Ww, device="cuda", requires_grad=True)
    g = torch.randn(B, C, Hh, Ww, device="cuda")
    out, ld = unit(x)
    assert ld == 0.0
    out.backward(g)
    dx_f = x.grad.clone()
    dws_f = [l.weight_fwd.grad.clone() for l in unit._chain()]
    # layer by layer
    x2 = x.detach().clone().requires_grad_(True)
    for l in unit._chain():
        l.weight_fwd.grad = None
    h = x2
    for l in unit._chain():
        h, _ = l(h)
    h.backward(g)
    assert torch.equal(out, h)
    assert torch.equal(dx_f, x2.grad)
    for a, l in zip(dws_f, unit._chain()):
        assert torch.equal(a, l.weight_fwd.grad)
    # oracle chain (fp64)
    z = x.detach().cpu().double().numpy()
    for l in unit._chain():
        z = O.inverse(z, l.weight_fwd.detach().cpu().double().numpy(), 0, l.order, nthreads=8)
    rel = float(np.linalg.norm(out.detach().cpu().double().numpy() - z) / np.linalg.norm(z))
    assert rel < 1e-5
    # round trip through reverse: four layers whose init (identity on the center tap, inv_conv.py:153-165) grows z along
    # each sweep, so the fp32 round trip carries the chain's conditioning (1.8e-3 at C=64, 32x32)
    back = unit.reverse(out.detach())
    assert float((back - x.detach()).norm() / x.detach().norm()) < 1e-2


@pytest.mark.gpu
def test_layers_inside_autocast(H):
    """config 2 trains under bf16 autocast: the inverse-conv layer and the block keep their arithmetic in fp32 there
    (custom_fwd(cast_inputs=float32)) and give the same result as outside the region."""
    from inf.layers.inv_conv import inv_flow_with_pad
    from inf.layers.inv_flow import Inv_FlowUnit
    torch.manual_seed(5)
    layer = inv_flow_with_pad(32, 32, (3, 3), order="TR").cuda()
    unit = Inv_FlowUnit(32, 32, (3, 3)).cuda()
    x = torch.randn(2, 32, 16, 16, device="cuda")
    ref_l, _ = layer(x)
    ref_u, _ = unit(x)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        xb = (x * 1.0)  # stays fp32; a bf16 producer upstream is cast back on entry
        out_l, _ = layer(xb)
        out_u, _ = unit(xb.to(torch.bfloat16).float())
        out_b, _ = layer(x.to(torch.bfloat16))
    assert out_l.dtype == torch.float32 and torch.equal(out_l, ref_l)
    assert out_b.dtype == torch.float32
    ref_b, _ = layer(x.to(torch.bfloat16).float())
    assert torch.equal(out_b, ref_b)
    ref_u2, _ = unit(x.to(torch.bfloat16).float())
    assert torch.equal(out_u, ref_u2)


def test_mini_glow_stack_end_to_end():
    """A small Glow level built like inf/if_multiGPU_imagenet32.py / if_glow_cifar.py (Squeeze, then blocks of ActNorm,
    inverse-flow layer, activation, Coupling) from this package's layers in a FlowSequential: every layer on the HIP
    library.  Checks: reverse(forward(x)) == x; the log-likelihood's directional derivative along a random direction
    of ALL parameters matches a central difference of the model's own forward; the input gradient likewise."""
    from inf.layers.actnorm import ActNorm
    from inf.layers.activations import SmoothLeakyRelu, SplineActivation
    from inf.layers.coupling import Coupling
    from inf.layers.flowsequential import FlowSequential
    from inf.layers.inv_conv import inv_flow_with_pad
    from inf.layers.squeeze import Squeeze
    from inf.train.losses import NegativeGaussianLoss
    torch.manual_seed(3)
    B, C, Hh, Ww = 4, 3, 8, 8
    size = (4 * C, Hh // 2, Ww // 2)
    layers = [Squeeze()]
    for k, order in enumerate(["TL", "BR"]):
        layers += [ActNorm(size[0]), inv_flow_with_pad(size[0], size[0], (3, 3), order=order),
                   SplineActivation(size, n_bins=5, tail_bound=4.0) if k == 0 else SmoothLeakyRelu(0.3),
                   Coupling(size, width=16)]
    model = FlowSequential(NegativeGaussianLoss(size=size), *layers).cuda()
    x = torch.randn(B, C, Hh, Ww, device="cuda")
    with torch.no_grad():
        model(x)  # data-dependent ActNorm initialisation
        for m in model.modules():  # give the zero-initialised coupling heads something to say
            if isinstance(m, Coupling):
                for p in m.net.parameters():
                    p.add_(0.05 * torch.randn_like(p))
    params = [p for p in model.parameters() if p.requires_grad]

    def nll(inp):
        z, lp = model(inp)
        return -(lp.sum() / B), z

    xg = x.clone().requires_grad_(True)
    loss, z = nll(xg)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    # reverse of the forward
    with torch.no_grad():
        h = z.detach()
        for m in reversed(list(model.sequence_modules)):
            h = m.reverse(h)
        assert float((h - x).abs().max()) < 2e-4
    # directional derivatives, one parameter tensor (and the input) at a time, against central differences of the model's
    # own forward in float32 (eps 5e-4: at 2e-3 the inverse-conv weights are already outside their linear range and the
    # conditioners cross ReLU kinks; all tensors at once do too)
    eps = 5e-4
    named = [(n_, p) for n_, p in model.named_parameters() if p.requires_grad] + [("input", None)]
    for name, p in named:
        d = torch.randn_like(x if p is None else p)
        with torch.no_grad():
            if p is None:
                analytic = float((xg.grad * d).sum())
                lp_, lm_ = nll(x + eps * d)[0], nll(x - eps * d)[0]
            else:
                analytic = float((p.grad * d).sum())
                p.add_(eps * d)
                lp_ = nll(x)[0]
                p.sub_(2 * eps * d)
                lm_ = nll(x)[0]
                p.add_(eps * d)
        numeric = float(lp_ - lm_) / (2 * eps)
        assert abs(analytic - numeric) < 3e-2 * abs(numeric) + 0.3, (name, analytic, numeric)


def test_level_captures_into_graphs():
    """Every launch of the library is stream-ordered and allocation-free, so a FlowSequential of its layers goes through
    torch.cuda.make_graphed_callables (forward and backward graphs); replays give the eager model's loss and gradients."""
    import copy
    from inf.layers.activations import SmoothLeakyRelu, SplineActivation
    from inf.layers.coupling import Coupling
    from inf.layers.flowsequential import FlowSequential
    from inf.layers.inv_conv import inv_flow_with_pad
    from inf.train.losses import NegativeGaussianLoss
    torch.manual_seed(11)
    B, size = 8, (12, 8, 8)
    layers = []
    for order in ("TL", "BR"):
        layers += [inv_flow_with_pad(12, 12, (3, 3), order=order), SplineActivation(size, tail_bound=4.0), SmoothLeakyRelu(0.3),
                   Coupling(size, width=16)]
    eager = FlowSequential(NegativeGaussianLoss(size=size), *layers).cuda()
    with torch.no_grad():
        for m in eager.modules():
            if isinstance(m, Coupling):
                for p in m.net.parameters():
                    p.add_(0.05 * torch.randn_like(p))
    graphed_model = copy.deepcopy(eager)
    x = torch.randn(B, *size, device="cuda")
    xg = x.clone().requires_grad_(True)
    graphed = torch.cuda.make_graphed_callables(graphed_model, (xg,))  # (before any eager backward of ITS parameters)
    for it in range(3):  # replays
        for p in graphed_model.parameters():
            p.grad = None
        xg.grad = None
        z, lp = graphed(xg)
        (-(lp.sum() / B)).backward()
    xe = x.clone().requires_grad_(True)
    ze, lpe = eager(xe)
    (-(lpe.sum() / B)).backward()
    assert rel_err(z.detach().cpu().numpy(), ze.detach().cpu().numpy()) < 1e-6
    assert rel_err(lp.detach().cpu().numpy(), lpe.detach().cpu().numpy()) < 1e-6
    assert rel_err(xg.grad.cpu().numpy(), xe.grad.cpu().numpy()) < 1e-5
    for (k, p), (_, q) in zip(graphed_model.named_parameters(), eager.named_parameters()):
        assert rel_err(p.grad.cpu().numpy(), q.grad.cpu().numpy()) < 1e-5, k
