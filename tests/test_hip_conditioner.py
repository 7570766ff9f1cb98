"""GPU: the fused conditioner of the affine coupling (csrc/conditioner.hip, ifl_cond_*) against the layer's own module tree --
the reference's `Coupling.net` (inf/layers/coupling.py:47-62: 3x3 conv, ReLU, 1x1 conv, ReLU, Conv2dZero) on library
convolutions, evaluated in fp64 (MIOpen's own fp32 kernels are off by up to 3e-3 in a weight gradient at these shapes:
tools/cond_err_probe.py).  Forward: 2e-5 of the output's scale.  Backward of an fp32 step: fp32 operand matrices for the
weight-gradient GEMMs, 2e-5 of each gradient's largest entry; under bf16 autocast the operand matrices are bf16 (fp32
accumulate -- the step's precision): 2e-2; d logs / d bias are fp32 sums either way: 1e-4."""
import copy
import pytest
import torch

import conftest  # noqa: F401

pytestmark = pytest.mark.gpu


def _coupling(C, width, seed):
    from inf.layers.coupling import Coupling
    torch.manual_seed(seed)
    layer = Coupling((C, 1, 1), width=width).cuda()
    c3 = layer.net[4]
    with torch.no_grad():  # (zero-initialised in the reference: give every parameter a value)
        c3.weight.normal_(0, 0.05)
        c3.bias = torch.nn.Parameter(torch.randn(C, device="cuda") * 0.1)
        c3.logs = torch.nn.Parameter(torch.randn(C, device="cuda") * 0.1)
    return layer


def _run(layer, x, gy, gld, fused, autocast=False):
    from inf.layers.coupling import Coupling
    Coupling.fused = fused
    try:
        for p in layer.parameters():
            p.grad = None
        xin = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y, ld = layer(xin)
        (y * gy).sum().add((ld * gld).sum()).backward()
        c1, c2, c3 = layer.net[0], layer.net[2], layer.net[4]
        return y.detach(), ld.detach(), xin.grad, [p.grad.clone() for p in (c1.weight, c2.weight, c3.weight, c3.bias, c3.logs)]
    finally:
        Coupling.fused = True


def _run64(layer, x, gy, gld):
    """the reference's Coupling.forward (coupling.py:66-88) on the same parameters in fp64"""
    l64 = copy.deepcopy(layer).double()
    xin = x.double().clone().requires_grad_()
    x1, x2, log_s, t = l64.get_xs_logs_t(xin)
    y, ld = torch.cat([x1, torch.addcmul(t, x2, log_s.exp())], dim=1), log_s.sum(dim=(1, 2, 3))
    (y * gy.double()).sum().add((ld * gld.double()).sum()).backward()
    c1, c2, c3 = l64.net[0], l64.net[2], l64.net[4]
    return y.detach(), ld.detach(), xin.grad, [p.grad for p in (c1.weight, c2.weight, c3.weight, c3.bias, c3.logs)]


SHAPES = [(100, 4, 14, 14, 512), (100, 8, 7, 7, 512), (32, 12, 16, 16, 128), (32, 24, 8, 8, 128), (13, 12, 16, 16, 256),
          (13, 24, 8, 8, 256), (13, 48, 4, 4, 256), (3, 16, 5, 3, 32), (1, 32, 2, 2, 16), (2, 4, 1, 9, 64), (5, 8, 16, 16, 48)]


@pytest.mark.parametrize("lowp", [False, True])
@pytest.mark.parametrize("B,C,H,W,width", SHAPES)
def test_fused_conditioner_matches_the_module_tree(B, C, H, W, width, lowp):
    import invflow_hip as Hh
    assert Hh.cond_supported(C, width)
    layer = _coupling(C, width, seed=B + C)
    torch.manual_seed(7)
    x = torch.randn(B, C, H, W, device="cuda")
    gy, gld = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, device="cuda")
    y1, ld1, gx1, gp1 = _run(layer, x, gy, gld, fused=True, autocast=lowp)  # (autocast only switches the GEMM operands)
    y0, ld0, gx0, gp0 = _run64(layer, x, gy, gld)
    tol = 2e-2 if lowp else 2e-5
    assert torch.allclose(y1.double(), y0, rtol=0, atol=2e-5 * max(1.0, float(y0.abs().max())))
    assert torch.allclose(ld1.double(), ld0, rtol=0, atol=2e-5 * max(1.0, float(ld0.abs().max())))
    assert float((gx1 - gx0).abs().max()) <= tol * float(gx0.abs().max())
    for name, a, b, tl in zip(("dW1", "dW2", "dW3", "db3", "dlogs"), gp1, gp0, (tol, tol, tol, 1e-4, 1e-4)):
        assert a.shape == b.shape, name
        assert float((a - b).abs().max()) <= tl * max(float(b.abs().max()), 1e-6), name
    # reverse of forward (sampling direction), no grad
    with torch.no_grad():
        assert torch.allclose(layer.reverse(y1), x, atol=1e-4 * max(1.0, float(x.abs().max())))


def test_fused_conditioner_is_reproducible_bit_for_bit():
    """no float atomics: two runs of forward + backward give identical bits (GEMMs included)"""
    layer = _coupling(12, 128, seed=3)
    x = torch.randn(32, 12, 16, 16, device="cuda")
    gy, gld = torch.randn_like(x), torch.randn(32, device="cuda")
    a = _run(layer, x, gy, gld, fused=True)
    b = _run(layer, x, gy, gld, fused=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))


def test_unsupported_shapes_keep_the_convolutions_and_bad_calls_fail_loudly():
    import invflow_hip as Hh
    from inf.layers.coupling import Coupling
    assert not Hh.cond_supported(6, 128) and not Hh.cond_supported(12, 100)
    layer = Coupling((6, 1, 1), width=64).cuda()
    x = torch.randn(2, 6, 4, 4, device="cuda")
    assert not layer._fusable(x, None)
    y, ld = layer(x)  # zero-initialised last convolution: the identity
    assert torch.allclose(y.detach(), x) and float(ld.detach().abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="not one of"):
        z = torch.zeros(8, device="cuda")
        Hh.cond_forward(x, z, z, z, None, 6, 64)


@pytest.mark.gpu
def test_weight_images_of_many_couplings_in_one_launch():
    """ifl_cond_prep_many_f32 (the job table of invflow_hip.CondPrepTable) fills the weight images of couplings of different
    shapes in one launch: every image identical, bit for bit, to ifl_cond_prep_f32's; a refill after the kernels changed
    follows them; a parameter that moved makes the table stale; an unsupported shape is refused when the table is built."""
    import invflow_hip as H
    torch.manual_seed(3)
    ents = []
    for C, width in [(12, 128), (24, 128), (4, 512), (48, 256), (8, 64)]:
        ents.append((torch.randn(width, C // 2, 3, 3, device="cuda"), torch.randn(C, width, 1, 1, device="cuda"),
                     torch.randn(C, C, 3, 3, device="cuda"), torch.randn(C, 1, 1, device="cuda") * 0.1, 3.0))
    table = H.CondPrepTable(ents)
    for round_ in range(2):
        table.run()
        for (w1, w2, w3, logs, f), wt in zip(ents, table.wt):
            assert torch.equal(wt, H.cond_prep(w1, w2, w3, logs, f))
        for e in ents:
            e[0].mul_(1.5); e[3].add_(0.01)  # (in place: the addresses stay)
    assert not table.stale(ents)
    moved = list(ents)
    moved[2] = (ents[2][0].clone(),) + ents[2][1:]
    assert table.stale(moved) and table.stale(ents[:-1])
    with pytest.raises(RuntimeError):
        H.CondPrepTable([(torch.randn(128, 5, 3, 3, device="cuda"), torch.randn(10, 128, 1, 1, device="cuda"),
                          torch.randn(10, 10, 3, 3, device="cuda"), torch.zeros(10, 1, 1, device="cuda"), 3.0)])
    assert H.CondPrepTable([]).n == 0
