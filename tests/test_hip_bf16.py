"""GPU: the bf16 storage variants of the C ABI (dtype row "bf16 storage / fp32 accumulate"; the reference
dispatches on the tensor's dtype, inv_conv_with_bp_kernel_general.cu:112).

Contract (include/invflow.h): a *_bf16 call returns exactly the round-to-nearest-even bf16 of what the *_f32 call returns on
the widened inputs; fp32 outputs (weight gradients, log-determinants, parameter gradients) are bit-identical.  So every
test here compares bit patterns with the f32 entry point, whose parity with the oracle the other GPU tests hold; one test
per family also checks against the oracle directly, within bf16's half-ulp."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def H():
    import invflow_hip
    invflow_hip.lib()
    return invflow_hip


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def layer_weights(C, K, seed, scale=0.05):
    g = torch.Generator().manual_seed(seed)
    w = scale * torch.randn(C, C, K, K, generator=g)
    w[:, :, -1, -1] = torch.tril(w[:, :, -1, -1], -1) + torch.eye(C)
    return w.cuda()


def bits(t):
    return t.view(torch.int16) if t.dtype == BF else t.view(torch.int32)


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and bool((bits(a) == bits(b)).all())


SHAPES = [(2, 64, 32, 32, 3, "TL"), (3, 8, 6, 6, 3, "BR"), (2, 32, 16, 16, 2, "TR"), (2, 256, 8, 8, 3, "TL"),
          (5, 4, 14, 14, 2, "BL"), (1, 3, 5, 7, 3, "TL")]


@pytest.mark.parametrize("B,C,HH,WW,K,order", SHAPES, ids=lambda v: str(v))
def test_layer_entry_points_round_the_f32_results(H, B, C, HH, WW, K, order):
    torch.manual_seed(B * 1000 + C)
    w = layer_weights(C, K, C + K)
    x = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    g = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    for with_carry in (False, True):
        c16 = H.new_carry(w) if with_carry else None
        c32 = H.new_carry(w) if with_carry else None
        z16 = H.inverse(x, w, order, carry=c16)
        z32 = H.inverse(x.float(), w, order, carry=c32)
        assert z16.dtype == BF and same_bits(z16, z32.to(BF))
        # the backward sees the STORED z: bf16
        dx16, dw16, _ = H.backward(g, z16, w, order, carry=c16)
        dx32, dw32, _ = H.backward(g.float(), z16.float(), w, order, carry=c32)
        assert dx16.dtype == BF and same_bits(dx16, dx32.to(BF))
        assert dw16.dtype == torch.float32 and same_bits(dw16, dw32)
    xh16, ld16 = H.forward(z16, w, order, want_logdet=True)
    xh32, ld32 = H.forward(z16.float(), w, order, want_logdet=True)
    assert xh16.dtype == BF and same_bits(xh16, xh32.to(BF)) and same_bits(ld16, ld32)
    # weights-only and input-gradient-only calls
    _, dw_only, _ = H.backward(g, z16, w, order, need_dx=False)
    assert same_bits(dw_only, dw32)
    dx_only, _, _ = H.backward(g, None, w, order, need_dw=False)
    assert same_bits(dx_only, dx32.to(BF))


def test_recon_term_in_bf16_storage(H):
    B, C, HH, WW, K = 4, 64, 32, 32, 3
    torch.manual_seed(3)
    w = layer_weights(C, K, 11)
    x = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    g = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    z = H.inverse(x, w)
    dx16, dw16, rl16 = H.backward(g, z, w, x=x, recon_weight=0.5)
    dx32, dw32, rl32 = H.backward(g.float(), z.float(), w, x=x.float(), recon_weight=0.5)
    assert same_bits(dx16, dx32.to(BF)) and same_bits(dw16, dw32)
    # (the loss is summed with one atomic per wave: the same value up to the order of the additions)
    assert abs(float(rl16) - float(rl32)) <= 1e-5 * abs(float(rl32))


ORACLE_SHAPES = [(2, 16, 12, 12, 3, "TL"), (2, 64, 32, 32, 3, "BR"), (2, 32, 16, 16, 2, "TR"), (1, 256, 8, 8, 3, "BL"),
                 (3, 4, 14, 14, 2, "TL"), (2, 12, 16, 16, 3, "TR"), (2, 7, 5, 9, 3, "BR")]


@pytest.mark.parametrize("B,C,HH,WW,K,order", ORACLE_SHAPES, ids=lambda v: str(v))
def test_against_the_oracle_within_half_an_ulp(H, oracle, B, C, HH, WW, K, order):
    """bf16 storage against the fp64 oracle itself (not against the f32 entry points), one shape per kernel family -- the
    two-workgroup MFMA scan, the whole-image MFMA scan with padded channels, the LDS-resident small layers, the wide team
    scan, the general kernel -- in every orientation: z, dx within half an ulp of bf16 plus the f32 path's 1e-5, dW (fp32
    storage) at the f32 path's tolerance on the widened operands."""
    torch.manual_seed(5)
    w = layer_weights(C, K, 2, scale=0.05 if C <= 16 else (0.02 if C <= 64 else 0.01))  # (well-conditioned: |z| stays O(|x|))
    x = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    g = torch.randn(B, C, HH, WW, device="cuda").to(BF)
    w64 = w.cpu().numpy().astype(np.float64)
    z = H.inverse(x, w, order)
    z_o = oracle.inverse(x.float().cpu().numpy().astype(np.float64), w64, order=order)
    half_ulp = lambda ref: 2.0 ** -8 * np.abs(ref) * (1 + 1e-3) + 1e-5 * np.abs(ref).max()
    assert np.all(np.abs(z.float().cpu().numpy() - z_o) <= half_ulp(z_o))
    # the backward of the STORED z (what the next call sees), against the oracle on the same widened operands
    zs = z.float().cpu().numpy().astype(np.float64)
    dx, dw, _ = H.backward(g, z, w, order)
    dx_o = oracle.dy(g.float().cpu().numpy().astype(np.float64), w64, order=order)
    dw_o = oracle.dw(zs, dx_o, (K, K), order=order)
    assert np.all(np.abs(dx.float().cpu().numpy() - dx_o) <= half_ulp(dx_o))
    assert dw.dtype == torch.float32 and rel_err(dw.cpu().numpy(), dw_o) < 3e-5
    # the layer's reverse (x^ = A z) of the stored z, against the oracle's product of the same z
    xh = H.forward(z, w, order)
    xh_o = oracle.forward(zs, w64, order=order)
    # (x^ is a sum of terms of the size of |z|: the f32 path's 1e-5 is relative to those)
    slack = 2.0 ** -8 * np.abs(xh_o) * (1 + 1e-3) + 1e-5 * max(np.abs(xh_o).max(), np.abs(zs).max())
    excess = np.abs(xh.float().cpu().numpy() - xh_o) - slack
    assert excess.max() <= 0, (excess.max(), np.abs(xh_o).max(), np.abs(zs).max())


def test_workspace_and_argument_checks(H):
    L = H.lib()
    w = layer_weights(8, 3, 1)
    x = torch.randn(2, 8, 6, 6, device="cuda").to(BF)
    z = torch.empty_like(x)
    need = L.ifl_workspace_bytes_bf16(H.OP_INVERSE, 2, 8, 6, 6, 3, 3, 0)
    assert need >= L.ifl_workspace_bytes(H.OP_INVERSE, 2, 8, 6, 6, 3, 3, 0) + 2 * x.numel() * 4
    small = torch.empty(need - 1, dtype=torch.uint8, device="cuda")
    rc = L.ifl_inverse_bf16(x.data_ptr(), w.data_ptr(), z.data_ptr(), 2, 8, 6, 6, 3, 3, 0, 0, small.data_ptr(), need - 1, None, None,
                            None)
    assert rc != 0 and b"workspace" in L.ifl_last_error()
    rc = L.ifl_inverse_bf16(None, w.data_ptr(), z.data_ptr(), 2, 8, 6, 6, 3, 3, 0, 0, small.data_ptr(), need - 1, None, None, None)
    assert rc != 0
    # empty batch: nothing to do, no workspace needed
    assert L.ifl_inverse_bf16(None, w.data_ptr(), None, 0, 8, 6, 6, 3, 3, 0, 0, None, 0, None, None, None) == 0
    with pytest.raises(RuntimeError):
        H.inverse(x, w, out=torch.empty(2, 8, 6, 6, device="cuda"))  # output must have the input's storage format


NEIGHBOUR_SHAPES = [(4, 8, 16, 16), (3, 6, 7, 7), (2, 12, 6, 10), (5, 2, 3, 5)]


@pytest.mark.parametrize("shape", NEIGHBOUR_SHAPES, ids=str)
def test_actnorm_bf16(H, shape):
    B, C, HH, WW = shape
    torch.manual_seed(C)
    x = torch.randn(*shape, device="cuda").to(BF)
    t, ls = torch.randn(C, device="cuda"), 0.3 * torch.randn(C, device="cuda")
    y16, ld16 = H.actnorm(x, t, ls)
    y32, ld32 = H.actnorm(x.float(), t, ls)
    assert same_bits(y16, y32.to(BF)) and same_bits(ld16, ld32)
    back = H.actnorm(y16, t, ls, reverse=True)
    assert same_bits(back, H.actnorm(y16.float(), t, ls, reverse=True).to(BF))
    gy, gld = torch.randn(*shape, device="cuda").to(BF), torch.randn(B, device="cuda")
    gx16, gt16, gls16 = H.actnorm_backward(gy, gld, x, t, ls)
    gx32, gt32, gls32 = H.actnorm_backward(gy.float(), gld, x.float(), t, ls)
    assert same_bits(gx16, gx32.to(BF)) and same_bits(gt16, gt32) and same_bits(gls16, gls32)


@pytest.mark.parametrize("shape", [(4, 3, 16, 16), (2, 5, 6, 10), (3, 1, 2, 2)], ids=str)
def test_squeeze_bf16_is_the_same_permutation(H, shape):
    x = torch.randn(*shape, device="cuda").to(BF)
    y = H.space_to_depth(x)
    assert y.dtype == BF and same_bits(y, torch.nn.functional.pixel_unshuffle(x, 2).contiguous())
    assert same_bits(H.depth_to_space(y), x)


@pytest.mark.parametrize("shape", NEIGHBOUR_SHAPES, ids=str)
def test_coupling_bf16(H, shape):
    B, C, HH, WW = shape
    torch.manual_seed(HH)
    x, h = torch.randn(*shape, device="cuda").to(BF), torch.randn(*shape, device="cuda").to(BF)
    y16, ld16 = H.coupling(x, h)
    y32, ld32 = H.coupling(x.float(), h.float())
    assert same_bits(y16, y32.to(BF)) and same_bits(ld16, ld32)
    assert same_bits(H.coupling(y16, h, reverse=True), H.coupling(y16.float(), h.float(), reverse=True).to(BF))
    gy, gld = torch.randn(*shape, device="cuda").to(BF), torch.randn(B, device="cuda")
    gx16, gh16 = H.coupling_backward(gy, gld, x, h)
    gx32, gh32 = H.coupling_backward(gy.float(), gld, x.float(), h.float())
    assert same_bits(gx16, gx32.to(BF)) and same_bits(gh16, gh32.to(BF))


def test_rounding_is_to_nearest_even_and_keeps_specials(H):
    # through ActNorm with t = 0, ls = 0 (y = x exactly in fp32): the store is the only rounding
    vals = torch.tensor([1.0, 1.00390625, 1.005859375, 1.001953125, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -2.5, 0.0, -0.0,
                         3.3895314e38, float("inf"), -float("inf"), 3.4e38, 65504.0, 1.17549435e-38, 7.0])
    x32 = vals.view(1, 1, 4, 4).cuda()
    t, ls = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    # (the f32 call is exact here; torch's own conversion is round-to-nearest-even)
    y32, _ = H.actnorm(x32, t, ls)
    assert same_bits(y32, x32)
    y16, _ = H.actnorm(x32.to(BF), t, ls)
    assert same_bits(y16, x32.to(BF))
    # narrowing inside the library: inverse with the identity layer (w = unit diagonal tap only) returns x
    w = torch.zeros(1, 1, 3, 3, device="cuda")
    w[0, 0, -1, -1] = 1.0
    xb = x32.to(BF)
    finite = torch.where(torch.isfinite(xb.float()), xb, torch.ones_like(xb))  # (inf times a zero tap would be NaN)
    z = H.inverse(finite, w)
    assert same_bits(z, H.inverse(finite.float(), w).to(BF))
    assert bool((z.float() == finite.float()).all())  # (the sign of a zero aside: x - 0 is computed as x + 0)


def test_host_layers_keep_bf16_activations(H):
    from inf.layers.actnorm import ActNorm
    from inf.layers.squeeze import Squeeze
    from inf.layers.inv_conv import inv_flow_with_pad
    torch.manual_seed(0)
    C = 8
    layer = inv_flow_with_pad(C, C, (3, 3)).cuda()
    with torch.no_grad():
        layer.weight_fwd.add_(0.05 * torch.randn_like(layer.weight_fwd) * layer.get_mask().to(layer.weight_fwd.device))
    x16 = torch.randn(3, C, 6, 6, device="cuda").to(BF).requires_grad_(True)
    x32 = x16.detach().float().requires_grad_(True)
    z16, _ = layer(x16)
    z16.float().square().sum().backward()
    gw16 = layer.weight_fwd.grad.clone()
    layer.weight_fwd.grad = None
    z32, _ = layer(x32)
    assert z16.dtype == BF and same_bits(z16.detach(), z32.detach().to(BF))
    assert x16.grad.dtype == BF and gw16.dtype == torch.float32
    # the f32 layer fed with the same (bf16-valued) upstream gradient and the STORED output gives the same gradients
    g = (2.0 * z16.detach().float()).to(BF)
    dx32, dw32, _ = H.backward(g.float(), z16.detach().float(), layer.weight_fwd.detach().contiguous(), layer.order)
    assert same_bits(x16.grad, dx32.to(BF))
    assert same_bits(gw16, dw32)
    # ActNorm and Squeeze modules
    an = ActNorm(C).cuda()
    an.initialized.fill_(1)
    y, ld = an(x16.detach())
    assert y.dtype == BF and ld.dtype == torch.float32
    with torch.no_grad():  # (the library's reverse is the sampling path: no autograd graph)
        assert same_bits(an.reverse(y), H.actnorm(y.detach(), an.translation.detach(), an.log_scale.detach(), reverse=True))
    sq = Squeeze()
    s, _ = sq(x16.detach())
    assert s.dtype == BF and same_bits(sq.reverse(s), x16.detach())


def test_bf16_step_replays_as_a_graph(H):
    """The bf16 entry points are stream-ordered and allocation-free like the f32 ones (the staging lives in the caller's
    workspace): a captured inverse + backward replays on new inputs and returns what the eager calls return."""
    torch.manual_seed(12)
    B, C, HH, WW, K = 4, 64, 32, 32, 3
    w = layer_weights(C, K, 5)
    xs = [torch.randn(B, C, HH, WW, device="cuda").to(BF) for _ in range(2)]
    gs = [torch.randn(B, C, HH, WW, device="cuda").to(BF) for _ in range(2)]
    eager = []
    for x, g in zip(xs, gs):
        z = H.inverse(x, w)
        dx, dw, _ = H.backward(g, z, w)
        eager.append((z, dx, dw))
    s = torch.cuda.Stream()
    xin, gin = xs[0].clone(), gs[0].clone()
    zg, dxg, dwg = torch.empty_like(xin), torch.empty_like(xin), torch.empty_like(w)
    carry = H.new_carry(w)
    with torch.cuda.stream(s):
        H.inverse(xin, w, out=zg, carry=carry)
        H.backward(gin, zg, w, dx_out=dxg, dw_out=dwg, carry=carry)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            H.inverse(xin, w, out=zg, carry=carry)
            H.backward(gin, zg, w, dx_out=dxg, dw_out=dwg, carry=carry)
    for k in (1, 0, 1):
        xin.copy_(xs[k])
        gin.copy_(gs[k])
        zg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert same_bits(zg, eager[k][0]) and same_bits(dxg, eager[k][1]) and same_bits(dwg, eager[k][2]), k
