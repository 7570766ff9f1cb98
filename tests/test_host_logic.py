"""CPU: host-side logic of the `inf` mirror package -- parameter init, masks, FlowSequential
bookkeeping -- none of which touches the GPU."""
import numpy as np
import pytest
import torch

import conftest  # noqa: F401  (sys.path)
from inf.layers.flowlayer import FlowLayer, ModifiedGradFlowLayer, PreprocessingFlowLayer, mark_expensive
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import flip_kernel, inv_flow_no_pad, inv_flow_with_pad
from inf.layers.inv_flow import Inv_FlowUnit
from inf.layers.selfnorm import SelfNormConv, SelfNormFC, _compute_weight_multiple
from inf.train.losses import NegativeGaussianLoss


def test_abcs():
    with pytest.raises(TypeError):
        FlowLayer()
    assert issubclass(ModifiedGradFlowLayer, FlowLayer) and issubclass(PreprocessingFlowLayer, FlowLayer)

    @mark_expensive
    def f():
        pass

    assert f._expensive_computation is True


@pytest.mark.parametrize("order", ["TL", "TR", "BL", "BR"])
def test_init_and_mask(order, oracle):
    torch.manual_seed(0)
    C, K = 6, 3
    layer = inv_flow_with_pad(C, C, (K, K), order=order, reference_init=True)
    w = layer.weight_fwd.detach()
    assert w.shape == (C, C, K, K) and layer.weight_fwd.requires_grad
    assert list(layer.state_dict().keys()) == ["weight_fwd"]  # checkpoint key of the reference (inv_conv.py:165)
    # reference recipe (inv_conv.py:153-179): identity at the kernel centre, W[c,-1,-1,-1]=1 before the order flip
    dims = {"TL": [], "TR": [3], "BL": [2], "BR": [2, 3]}[order]
    wt = torch.flip(w, dims) if dims else w
    assert torch.all(wt[:, -1, -1, -1] == 1.0)
    assert torch.allclose(wt[torch.arange(C - 1), torch.arange(C - 1), K // 2, K // 2], torch.ones(C - 1), atol=0.05)
    # the default: the identity at the operator's diagonal tap (the layer starts as the identity map; README deviation 9) --
    # same shape, same key, nothing but noise at the centre
    safe = inv_flow_with_pad(C, C, (K, K), order=order)
    ws = torch.flip(safe.weight_fwd.detach(), dims) if dims else safe.weight_fwd.detach()
    assert ws.shape == w.shape and list(safe.state_dict().keys()) == ["weight_fwd"]
    assert torch.allclose(ws[torch.arange(C), torch.arange(C), -1, -1], torch.ones(C), atol=0.05)
    assert float(ws[:, :, K // 2, K // 2].abs().max()) < 0.05
    # mask = reference get_mask (inv_conv.py:233-248) = oracle mask
    m = layer.get_mask().numpy()
    assert np.array_equal(m, oracle.mask(C, K, K, 0, order))
    assert layer.pad == {"TL": (2, 0, 2, 0), "TR": (0, 2, 2, 0), "BL": (2, 0, 0, 2), "BR": (0, 2, 0, 2)}[order]
    # reset_gradients applies it
    layer.weight_fwd.grad = torch.ones_like(w)
    layer.reset_gradients()
    assert np.array_equal(layer.weight_fwd.grad.numpy(), m)
    assert layer.logdet(torch.zeros(3, C, 4, 4)) == 0.0
    assert torch.equal(layer.logdet(torch.zeros(3, C, 4, 4), compute_expensive=True), torch.zeros(3))


def test_no_pad_and_unit():
    layer = inv_flow_no_pad(4, 4, (3, 3))
    assert layer.order == "TL" and layer.kernel_size == (3, 3)
    with pytest.raises(AssertionError):
        inv_flow_with_pad(4, 4, (3, 3), order="XX")
    unit = Inv_FlowUnit(4, 4, 3)
    assert [l.order for l in unit._chain()] == ["TL", "TR", "BL", "BR"]
    assert len(list(unit.parameters())) == 4
    w = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).view(2, 3, 2, 2)
    fk = flip_kernel(w)
    assert fk.shape == (3, 2, 2, 2) and fk[1, 0, 0, 0] == w[0, 1, 1, 1]


class _Scale(FlowLayer):
    """CPU stand-in layer with a tensor log-det."""

    def __init__(self, s):
        super().__init__()
        self.s = s

    def forward(self, input, context=None):
        return input * self.s, torch.full((len(input),), float(np.log(abs(self.s)) * input[0].numel()))

    def reverse(self, input, context=None):
        return input / self.s

    def logdet(self, input, context=None):
        return self.forward(input)[1]


class _Shift(FlowLayer):
    """CPU stand-in layer with the python-float log-det 0.0 the inverse-conv layers return."""

    def forward(self, input, context=None):
        return input + 1.0, 0.0

    def reverse(self, input, context=None):
        return input - 1.0

    def logdet(self, input, context=None):
        return 0.0


def test_flowsequential_counts_logdet_once():
    base = NegativeGaussianLoss((2, 3, 3))
    model = FlowSequential(base, _Scale(2.0), _Shift(), _Scale(0.5), _Shift())
    x = torch.randn(4, 2, 3, 3)
    out, lp = model(x)
    expect = base.log_prob(out) + 0.0  # log 2 + log 0.5 = 0, counted once each
    assert torch.allclose(lp, expect, atol=1e-6)
    out2, lp2 = FlowSequential(base, _Scale(2.0))(x)
    assert torch.allclose(lp2 - base.log_prob(out2), torch.full((4,), float(np.log(2.0) * 18)), atol=1e-5)
    assert torch.allclose(model.reconstruct(x), x, atol=1e-6)
    s, st = model.sample(5)
    assert s.shape == (5, 2, 3, 3) and torch.equal(s, st)
    assert model.add_recon_grad() == 0.0 and list(model.selfnorm_modules()) == []
    assert len(list(model.non_preprocessing_modules())) == 4
    assert torch.allclose(model.log_prob(x), lp) and torch.allclose(model.cheap_unnormed_log_prob(x), lp)


def test_selfnorm_host_pieces():
    torch.manual_seed(1)
    conv = SelfNormConv(4, 4, (3, 3), padding=1)
    assert sorted(conv.state_dict()) == ["bias_fwd", "weight_fwd", "weight_inv"]
    assert torch.equal(conv.weight_inv.data, flip_kernel(conv.weight_fwd.data))
    with pytest.raises(NotImplementedError):
        SelfNormConv(4, 4, (3, 3), stride=2)
    fc = SelfNormFC(10, 10)
    assert fc.weight_fwd.shape == (10, 10, 1, 1)
    # closed-form weight multiple == conv2d_weight(ones, ones)/B  (selfnorm.py:24-32)
    x = torch.ones(2, 3, 6, 5)
    m = _compute_weight_multiple((4, 3, 3, 3), x, (1, 1))
    ref = torch.nn.grad.conv2d_weight(x, (4, 3, 3, 3), torch.ones(2, 4, 6, 5), 1, 1) / 2
    assert torch.allclose(m, ref)


def test_storage_format_of_an_activation_picks_the_entry_point():
    """invflow_hip._storage: bf16 activations -> the *_bf16 entry points, everything else -> *_f32 (whose dtype check then
    refuses what is not float32), as the reference dispatches on the tensor's dtype (inv_conv_with_bp_kernel_general.cu:112)."""
    import torch
    import invflow_hip as H
    assert H._storage(torch.zeros(1, dtype=torch.bfloat16), "x") == ("bf16", torch.bfloat16)
    assert H._storage(torch.zeros(1), "x") == ("f32", torch.float32)
    assert H._storage(torch.zeros(1, dtype=torch.float16), "x") == ("f32", torch.float32)
    for name in ("ifl_inverse", "ifl_forward", "ifl_backward", "ifl_actnorm", "ifl_actnorm_backward", "ifl_squeeze", "ifl_coupling",
                 "ifl_coupling_backward"):
        assert name + "_bf16" in H.SIGNATURES and name + "_f32" in H.SIGNATURES
        assert H.SIGNATURES[name + "_bf16"] == H.SIGNATURES[name + "_f32"]  # same argument lists: only the pointee differs
    with __import__("pytest").raises(RuntimeError):
        H.inverse(torch.zeros(1, 1, 2, 2, dtype=torch.bfloat16), torch.zeros(1, 1, 1, 1))  # CPU tensors: no fallback


def test_gradient_masks_are_applied_in_place_and_stay_in_the_bucket():
    """reset_gradients (inv_conv.py:223-230, conv.py:99-100) must not REPLACE .grad: the gradients are views of one flat bucket
    that is zeroed and all-reduced as a whole (data_parallel.GradBucket), and a captured step keeps their addresses.  After
    clear_grad every .grad still aliases the bucket, carries the mask, and bucket.zero() clears it."""
    import data_parallel as dp
    from inf.layers.conv import PaddedConv2d
    from inf.train.step import clear_grad
    torch.manual_seed(0)
    mods = torch.nn.ModuleList([inv_flow_with_pad(6, 6, (3, 3), order="TR"), inv_flow_with_pad(6, 6, (2, 2), order="BL"),
                                PaddedConv2d(6, 6, (3, 3), order="BR")])
    bucket = dp.GradBucket(mods.parameters())
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
    for rep in range(3):
        bucket.flat.normal_()
        before = [p.grad.clone() for p in mods.parameters()]
        mods.apply(clear_grad)
        mods[2].reset_gradients()
        for p in mods.parameters():
            assert lo <= p.grad.data_ptr() < hi
        for m, g0 in ((mods[0], before[0]), (mods[1], before[1])):
            assert torch.equal(m.weight_fwd.grad, g0 * m.get_mask()) and float((m.weight_fwd.grad == 0).sum()) > 0
        assert torch.equal(mods[2].conv.weight.grad, before[2] * mods[2].mask)
        bucket.zero()
        assert all(float(p.grad.abs().sum()) == 0.0 for p in mods.parameters())


def test_bucket_release_and_gather():
    """GradBucket.release() / gather() (TrainStep's captured backward writes fresh gradient tensors, one multi-tensor copy
    brings them into the flat bucket): the bucket holds the gradients in parameter order, a parameter that received none
    holds zeros, every .grad is a bucket view again -- and accumulating into the views afterwards lands in the bucket."""
    import data_parallel as dp
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2, 2))]
    bucket = dp.GradBucket(ps)
    bucket.flat.fill_(7.0)
    bucket.release()
    assert all(p.grad is None for p in ps)
    g0, g2 = torch.randn(3, 4), torch.randn(2, 2, 2)
    ps[0].grad, ps[2].grad = g0, g2  # (the second parameter gets none)
    bucket.gather()
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in ps)
    assert torch.equal(bucket.flat, torch.cat([g0.flatten(), torch.zeros(5), g2.flatten()]))
    ps[1].grad.add_(1.0)
    assert torch.equal(bucket.flat[12:17], torch.ones(5))
    bucket.release()
    for p, g in zip(ps, (g0, torch.ones(5), g2)):
        p.grad = g.clone()
    bucket.gather()  # (all present: no zero fill needed)
    assert torch.equal(bucket.flat, torch.cat([g0.flatten(), torch.ones(5), g2.flatten()]))

