"""Checkpoint compatibility: the package's model loads a state dict laid out by the reference's own
layer classes (tests/golden/state_dict_glow_L2K2.npz, written by tests/golden/make_golden_state_dict.py), and the
save/load pair keeps the reference's checkpoint layout (inf/train/experiment.py:475-502).  The FInC-flow convolution
(inf/layers/conv.py:22-222) reverses on the HIP library."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR, rel_err

CFG = dict(inv_flow=True, inv_conv_no_pad=True, if_kernel_size=3, coupling_width=16, num_blocks=2, block_size=2,
           tail_bound=20, n_bins=5, actnorm=True, activation="Spline", split_prior=True, image_size=(1, 8, 8),
           dequantize=True, split_width=16)


def reference_state_dict():
    d = np.load(os.path.join(GOLDEN_DIR, "state_dict_glow_L2K2.npz"))
    return {str(k): torch.from_numpy(d[str(k)]) for k in d["__keys__"]}


def test_reference_state_dict_loads_strict():
    from inf.experiments.if_glow_mnist import create_model
    model = create_model(**CFG)
    ref = reference_state_dict()
    assert list(model.state_dict().keys()) == list(ref.keys())  # same names in the same order
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(ref[k].shape) and v.dtype == ref[k].dtype, k
    model.load_state_dict(ref, strict=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref[k]), k
    assert any(k.endswith("weight_fwd") for k in ref)  # experiment.py:475-502 checkpoints carry `...weight_fwd`


def test_checkpoint_layout_round_trip(tmp_path):
    from inf.experiments.if_glow_mnist import create_model
    from inf.train.checkpoint import KEYS, load_checkpoint, save_checkpoint
    model = create_model(**CFG)
    model.load_state_dict(reference_state_dict())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    for p in model.parameters():  # one optimizer step so that the optimizer state is not empty
        p.grad = torch.full_like(p, 0.01)
    opt.step()
    sched.step()
    path = str(tmp_path / "checkpoint.tar")
    save_checkpoint(path, model, opt, sched, summary={"Epoch": 3}, config={"lr": 1e-3, "name": "IF_Glow"})
    raw = torch.load(path, weights_only=False)
    assert tuple(raw.keys()) == KEYS  # the reference's key order, experiment.py:477-483
    model2 = create_model(**CFG)
    opt2 = torch.optim.Adam(model2.parameters(), lr=1.0)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=1, gamma=0.5)
    summary, config = load_checkpoint(path, model2, opt2, sched2)
    assert summary == {"Epoch": 3} and config["name"] == "IF_Glow"
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] == 5e-4
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)
    with pytest.raises(KeyError):
        torch.save({"model_state_dict": model.state_dict()}, path)
        load_checkpoint(path, model2)


def test_padded_conv_surface():
    """conv.py:33-100: pads per order, `conv.weight` key, unit lower-triangular diagonal tap at the order's corner, mask"""
    from inf.layers.conv import Finc_FlowUnit, PaddedConv2d
    pads = {"TL": (2, 0, 2, 0), "TR": (0, 2, 2, 0), "BL": (2, 0, 0, 2), "BR": (0, 2, 0, 2)}
    corner = {"TL": (-1, -1), "TR": (-1, 0), "BL": (0, -1), "BR": (0, 0)}
    for order in pads:
        layer = PaddedConv2d(5, 5, (3, 3), order=order)
        assert layer.pad == pads[order]
        assert list(layer.state_dict().keys()) == ["conv.weight"]
        w = layer.conv.weight.data
        tap = w[:, :, corner[order][0], corner[order][1]]
        assert torch.equal(torch.diagonal(tap), torch.ones(5)) and torch.equal(torch.triu(tap, 1), torch.zeros(5, 5))
        m = layer.get_mask()
        mtap = m[:, :, corner[order][0], corner[order][1]]
        assert torch.equal(mtap, torch.tril(torch.ones(5, 5), -1)) and m.sum() == 5 * 5 * 9 - 15
        x = torch.randn(2, 5, 6, 4)
        y, ld = layer(x)
        assert y.shape == x.shape and ld == 0.0
        y.sum().backward()
        layer.reset_gradients()
        assert torch.equal(layer.conv.weight.grad * (1 - m), torch.zeros_like(m))
        with pytest.raises(RuntimeError, match="CUDA tensor"):
            layer.reverse(y.detach())
    unit = Finc_FlowUnit(8, 8, 3)
    assert sorted(unit.state_dict().keys()) == ["conv_bl.conv.weight", "conv_br.conv.weight", "conv_tl.conv.weight",
                                                 "conv_tr.conv.weight"]
    assert unit(torch.randn(1, 8, 5, 5))[0].shape == (1, 8, 5, 5)


def test_padded_conv_forward_is_the_oracle_operator(oracle):
    """forward = A x with A the operator the inverse undoes: padded torch conv == oracle.forward for every order"""
    from inf.layers.conv import PaddedConv2d
    torch.manual_seed(3)
    for order in ("TL", "TR", "BL", "BR"):
        for K in (2, 3):
            layer = PaddedConv2d(6, 6, (K, K), order=order).double()
            x = torch.randn(2, 6, 7, 5, dtype=torch.float64)
            y, _ = layer(x)
            y_o = oracle.forward(x.numpy(), layer.conv.weight.detach().numpy(), 0, order)
            assert rel_err(y.detach().numpy(), y_o) < 1e-12


# ---- on the GPU ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("order", ["TL", "TR", "BL", "BR"])
@pytest.mark.parametrize("shape", [(1, 4, 5, 5, 3), (2, 6, 7, 5, 2), (3, 64, 32, 32, 3), (2, 32, 16, 16, 2), (2, 12, 32, 32, 3)])
def test_padded_conv_reverse_on_hip(oracle, order, shape):
    """conv.py:102-165: reverse(forward(x)) == x (tests/inf/test_layers.py check_inverse, atol 1e-3) and reverse == the
    oracle's inverse of the same kernel"""
    import invflow_hip
    from inf.layers.conv import PaddedConv2d
    B, C, H, W, K = shape
    torch.manual_seed(hash((order, shape)) % 1000)
    invflow_hip.lib()  # (fails loudly when the HIP library is missing)
    layer = PaddedConv2d(C, C, (K, K), order=order).cuda()
    x = torch.randn(B, C, H, W, device="cuda")
    y, ld = layer(x)
    back, ld_r = layer.reverse(y.detach())
    assert ld == 0.0 and ld_r == 0
    if C <= 12:  # the reference's own criterion at its test sizes (tests/inf/test_layers.py:19-36)
        np.testing.assert_allclose(back.cpu().numpy(), x.cpu().numpy(), atol=1e-3)
    # N(0, 0.05) kernels over 64 x 9 taps amplify the fp32 rounding of the forward conv: relative criterion at size
    assert rel_err(back.cpu().numpy(), x.cpu().numpy()) < 1e-4
    y_h, w_h = y.detach().cpu().numpy(), layer.conv.weight.detach().cpu().numpy()
    z_o = oracle.inverse(y_h.astype(np.float64), w_h.astype(np.float64), 0, order, nthreads=8)
    # these kernels are badly conditioned at C = 64 (the fp32 solve of the reference loses digits too): the bound is the
    # larger of the usual 1e-5 and twice what the oracle's own fp32 run loses against its fp64 run on the same input
    z_o32 = oracle.inverse(y_h, w_h, 0, order, nthreads=8)
    assert z_o32.dtype == np.float32
    assert rel_err(back.cpu().double().numpy(), z_o) < max(1e-5, 2 * rel_err(z_o32, z_o))


@pytest.mark.gpu
def test_finc_unit_reverse_on_hip():
    from inf.layers.conv import Finc_FlowUnit
    torch.manual_seed(5)
    unit = Finc_FlowUnit(128, 128, (3, 3)).cuda()
    x = torch.randn(2, 128, 16, 16, device="cuda")
    y, ld = unit(x)
    assert ld == 0.0
    assert rel_err(unit.reverse(y.detach()).cpu().numpy(), x.cpu().numpy()) < 1e-4


@pytest.mark.gpu
def test_reference_checkpoint_runs_on_the_hip_layers(tmp_path):
    """A state dict in the reference's layout -> the package's model on the GPU: forward, save, load into a fresh model,
    identical log-likelihoods; the reconstruction through reverse() comes back to the input."""
    from inf.experiments.if_glow_mnist import create_model
    from inf.train.checkpoint import load_checkpoint, save_checkpoint
    cfg = dict(CFG, dequantize=False)  # (the dequantisation noise is random: leave it out to compare outputs)
    ref = {k: v for k, v in reference_state_dict().items() if not k.startswith("0.")}
    # without the Dequantization entry the module indices shift by one
    ref = {".".join([str(int(k.split(".")[0]) - 1)] + k.split(".")[1:]): v for k, v in ref.items()}
    model = create_model(**cfg).cuda()
    model.load_state_dict(ref, strict=True)
    model.eval()
    torch.manual_seed(1)
    x = torch.randint(0, 256, (6, 1, 8, 8), device="cuda").float() + 0.5
    with torch.no_grad():
        z, logp = model(x)
    assert torch.isfinite(logp).all()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    path = str(tmp_path / "checkpoint.tar")
    save_checkpoint(path, model, opt, sched, summary={"Epoch": 1}, config=cfg)
    model2 = create_model(**cfg).cuda()
    load_checkpoint(path, model2, map_location="cuda")
    model2.eval()
    with torch.no_grad():
        z2, logp2 = model2(x)
    assert torch.equal(z, z2) and torch.equal(logp, logp2)
