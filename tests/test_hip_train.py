"""GPU: the train-step replacement (inf/train/step.py: get_loss, backward, clip, gradient bucket, optimizer step --
inf/train/experiment.py:160-195,272-311) on the MNIST-Glow-shaped model built from this package's layers
(inf/experiments/if_glow_mnist.py), against a CPU fp64 run of the same model built from the reference's own layers
(tests/golden/make_golden_trainstep.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fixture():
    d = np.load(os.path.join(GOLDEN, "trainstep_glow_b6_8x8_L2K2.npz"))
    return {k: d[k] for k in d.files}


def build(fixture, dtype=torch.float32):
    from inf.experiments.if_glow_mnist import create_model
    B, c, h, w, width, nb, tb = (int(v) for v in fixture["config"])
    model = create_model(dequantize=False, image_size=(c, h, w), num_blocks=2, block_size=2, coupling_width=width,
                         split_width=width, n_bins=nb, tail_bound=tb, inv_conv_no_pad=True).cuda()
    sd = {k[3:]: torch.from_numpy(v).to(dtype) for k, v in fixture.items() if k.startswith("sd/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected  # the reference-layer model's state_dict loads as is
    assert all("mask" in k or "_device_probe" in k for k in missing), missing
    return model


def test_loss_and_every_gradient_match_reference_layers(fixture):
    from inf.train.step import get_loss
    model = build(fixture)
    x = torch.from_numpy(fixture["x"]).float().cuda()
    loss = get_loss(model, x)
    loss.backward()
    assert abs(float(loss.detach()) - float(fixture["loss"])) < 2e-5 * abs(float(fixture["loss"]))
    grads = {k[5:]: v for k, v in fixture.items() if k.startswith("grad/")}
    params = dict(model.named_parameters())
    assert set(grads) == set(params)
    worst = 0.0
    for name, g_ref in grads.items():
        g = params[name].grad
        assert g is not None, name
        if np.linalg.norm(g_ref) == 0:
            assert float(g.abs().max()) < 1e-6, name
            continue
        e = rel_err(g.detach().cpu().double().numpy(), g_ref)
        worst = max(worst, e)
        assert e < 5e-4, (name, e)  # fp32 through 20 layers against fp64
    assert worst > 0


def test_train_step_runs_and_learns(fixture):
    from inf.train.step import TrainStep, bits_per_dim
    torch.manual_seed(0)
    model = build(fixture)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    step = TrainStep(model, opt, grad_clip_norm=1.0, clear_grads=True)
    x = torch.from_numpy(fixture["x"]).float().cuda()
    # every gradient is a view of the one flat bucket the data-parallel all-reduce works on
    assert all(p.grad.data_ptr() >= step.bucket.flat.data_ptr() for p in model.parameters())
    losses = [float(step(x)) for _ in range(25)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 1.0
    assert abs(losses[0] - float(fixture["loss"])) < 2e-5 * abs(float(fixture["loss"]))
    assert bits_per_dim(losses[0], x[0].numel()) > 0
    # bf16 autocast around the model (BASELINE configs[2]): the library layers compute in fp32, the conditioners in bf16
    model2 = build(fixture)
    step2 = TrainStep(model2, torch.optim.Adam(model2.parameters(), lr=1e-3), autocast=True)
    l2 = float(step2(x))
    assert abs(l2 - float(fixture["loss"])) < 2e-2 * abs(float(fixture["loss"]))


@pytest.fixture(scope="module")
def cifar_fixture():
    d = np.load(os.path.join(GOLDEN, "trainstep_glow_cifar_b4_8x8_L2K2.npz"))
    return {k: d[k] for k in d.files}


def build_cifar(fixture):
    from inf.experiments.if_glow_cifar import create_model
    B, c, h, w, width, nb, tb = (int(v) for v in fixture["config"])
    model = create_model(dequantize=False, image_size=(c, h, w), num_blocks=2, block_size=2, coupling_width=width, split_width=width,
                         inv_flow=True, if_kernel_size=3, actnorm=True).cuda()
    sd = {k[3:]: torch.from_numpy(v).float() for k, v in fixture.items() if k.startswith("sd/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("mask" in k or "_device_probe" in k for k in missing), missing
    return model


def test_cifar_shaped_glow_matches_reference_layers(cifar_fixture):
    """The if_glow_cifar / if_glow_imagenet32 model shape (BASELINE configs[3], configs[4]): three colour channels, 3x3
    inverse-flow layers, one shared spline per step and none behind the last, against the fp64 run of the reference's own
    layers (tests/golden/make_golden_trainstep.py --cifar): loss and every parameter gradient; then the graph step."""
    from inf.train.step import TrainStep, get_loss
    fixture = cifar_fixture
    model = build_cifar(fixture)
    x = torch.from_numpy(fixture["x"]).float().cuda()
    loss = get_loss(model, x)
    loss.backward()
    assert abs(float(loss.detach()) - float(fixture["loss"])) < 2e-5 * abs(float(fixture["loss"]))
    grads = {k[5:]: v for k, v in fixture.items() if k.startswith("grad/")}
    params = dict(model.named_parameters())
    assert set(grads) == set(params)
    for name, g_ref in grads.items():
        g = params[name].grad
        assert g is not None, name
        if np.linalg.norm(g_ref) == 0:
            assert float(g.abs().max()) < 1e-6, name
            continue
        assert rel_err(g.detach().cpu().double().numpy(), g_ref) < 5e-4, name
    # sampling direction: the model's reverse undoes its forward
    with torch.no_grad():
        out, _ = model(x)
    model2 = build_cifar(fixture)
    step = TrainStep(model2, torch.optim.Adam(model2.parameters(), lr=1e-3), grad_clip_norm=1.0, clear_grads=True, graph=True,
                     graph_warmup=2)
    losses = [float(step(x)) for _ in range(8)]
    assert abs(losses[0] - float(fixture["loss"])) < 2e-5 * abs(float(fixture["loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] and step._captured is not None


def _steps_at_config_size(module_name, cfg, per_rank_batch, actnorm=None):
    """Three training steps of a model built exactly as its configuration says -- `actnorm` False for both 32x32x3
    configurations (if_glow_cifar.py:147, if_multiGPU_imagenet32.py:284-345) -- on uniform noise.  The layers start as the
    identity map (inf/layers/inv_conv.py _init_weight: the identity at the operator's diagonal tap; the reference's recipe
    puts it at the kernel centre, which for the exact operator is a one-pixel shift whose inverse amplifies ~6x per layer and
    takes a model of 32 such layers without ActNorm out of fp32 on its first batch: reference_init=True, exercised below
    WITH ActNorm).  actnorm=None: as configured."""
    import importlib
    from inf.train.step import TrainStep, bits_per_dim
    create_model = importlib.import_module(module_name).create_model
    torch.manual_seed(1)
    an = cfg["actnorm"] if actnorm is None else actnorm
    model = create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                         if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                         coupling_width=cfg["coupling_width"], n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"],
                         activation=cfg["activation"], actnorm=an, split_prior=cfg["split_prior"],
                         reference_init=(actnorm is True)).cuda()
    x = torch.randint(0, 256, (per_rank_batch, 3, 32, 32)).float().cuda()
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True)
    # (the variant with the reference's initialisation: two steps, as in round 2 -- 144 amplifying layers at the configured
    # learning rate of 1e-3 on noise do not survive many more)
    losses = [float(step(x)) for _ in range(3 if actnorm is None else 2)]
    assert all(np.isfinite(losses)), losses
    bpd = bits_per_dim(losses[0], 3 * 32 * 32)
    if actnorm is None:
        assert 2.0 < bpd < 12.0, bpd  # (uniform noise through an untrained model that starts as the identity: ~8 bits per dimension)
    else:
        assert 0.5 < bpd < 100.0, bpd
    # every inverse-flow layer of the model got a gradient through the library's fused backward
    from inf.layers.inv_conv import _InvFlowBase
    layers = [m for m in model.modules() if isinstance(m, _InvFlowBase)]
    assert len(layers) == cfg["num_blocks"] * cfg["block_size"]
    assert all(m.weight_fwd.grad is not None and bool(torch.isfinite(m.weight_fwd.grad).all()) for m in layers)
    return sorted({m.in_channels for m in layers})


def test_config4_cifar_glow_at_its_own_size():
    """if_glow_cifar AS CONFIGURED (if_glow_cifar.py:108-190: L = 2, K = 16, 3x3 inverse-flow layers, coupling width 128, no
    ActNorm) at BASELINE configs[3]'s per-GPU batch (256 over eight ranks): three training steps on synthetic data; then the
    variant with ActNorm and the reference's initialisation."""
    from inf.experiments.if_glow_cifar import DEFAULT_CONFIG
    assert DEFAULT_CONFIG["actnorm"] is False
    assert _steps_at_config_size("inf.experiments.if_glow_cifar", DEFAULT_CONFIG, 32) == [12, 24]
    assert _steps_at_config_size("inf.experiments.if_glow_cifar", DEFAULT_CONFIG, 32, actnorm=True) == [12, 24]


def test_config5_imagenet32_glow_at_its_own_size():
    """The multi-GPU ImageNet-32 model AS CONFIGURED (if_multiGPU_imagenet32.py:284-345: L = 3, K = 48, coupling width 256, no
    ActNorm): three training steps at a rank's shard of the batch of 100 over eight ranks; then the ActNorm variant."""
    from inf.experiments.if_glow_imagenet32 import DEFAULT_CONFIG
    assert DEFAULT_CONFIG["actnorm"] is False
    assert _steps_at_config_size("inf.experiments.if_glow_imagenet32", DEFAULT_CONFIG, 13) == [12, 24, 48]
    assert _steps_at_config_size("inf.experiments.if_glow_imagenet32", DEFAULT_CONFIG, 13, actnorm=True) == [12, 24, 48]


def test_config3_model_builds_at_its_own_size():
    """if_glow_mnist as configured (L = 2, K = 16, batch 100, 28x28x1, 2x2 inverse-flow layers, per-element splines,
    coupling width 512): one training step on synthetic dequantised data."""
    from inf.experiments.if_glow_mnist import DEFAULT_CONFIG, create_model
    from inf.train.step import TrainStep, bits_per_dim
    torch.manual_seed(1)
    cfg = DEFAULT_CONFIG
    model = create_model(num_blocks=cfg["num_blocks"], block_size=cfg["block_size"], coupling_width=cfg["coupling_width"],
                         n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"]).cuda()
    x = torch.randint(0, 256, (cfg["batch_size"], 1, 28, 28)).float().cuda()
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True)
    l0 = float(step(x))
    l1 = float(step(x))
    assert np.isfinite(l0) and np.isfinite(l1)
    assert 0.5 < bits_per_dim(l0, 28 * 28) < 20.0


def test_train_step_as_one_graph(fixture):
    """TrainStep(graph=True): after its eager warm-up the whole step (forward, backward, clip, Adam) is one captured graph;
    the loss sequence is that of the eager step on the same batches (no dequantisation noise in this model), and a batch of
    another shape is refused."""
    from inf.train.step import TrainStep
    x = torch.from_numpy(fixture["x"]).float().cuda()
    xs = [x, x * 0.5 + 64.0, x * 0.25 + 128.0]
    seqs = []
    for graph in (False, True):
        torch.manual_seed(0)
        model = build(fixture)
        step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), grad_clip_norm=1.0, clear_grads=True, graph=graph,
                         graph_warmup=2)
        seqs.append([float(step(xs[i % 3])) for i in range(9)])
        if graph:
            assert step._captured is not None
            with pytest.raises(ValueError):
                step(x[:3])
    np.testing.assert_allclose(seqs[1], seqs[0], rtol=2e-4)
    assert seqs[1][-1] < seqs[1][0]


def test_couplings_weight_images_prepared_in_one_launch_change_nothing():
    """TrainStep(batch_cond_prep=True, the default) prepares the weight images of all fused couplings in one launch at the head
    of the step (inf.layers.coupling.ConditionerPrep) instead of one launch inside every coupling's forward: the same
    arithmetic on the same numbers -- the loss sequences with and without it are identical, eagerly and as a captured graph
    (whose job table survives the re-homing of the parameters into the flat optimizer buffer); outside a step the layers
    hold no prepared image."""
    from inf.train.step import TrainStep
    from inf.layers.coupling import Coupling
    import inf.experiments.if_glow_cifar as mod
    cfg = mod.DEFAULT_CONFIG
    x = torch.randint(0, 256, (4, 3, 32, 32), device="cuda").float()
    seqs = {}
    for graph in (False, True):
        for batched in (False, True):
            torch.manual_seed(2)
            model = mod.create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                                     if_kernel_size=cfg["if_kernel_size"], num_blocks=2, block_size=2, coupling_width=32,
                                     activation=cfg["activation"], actnorm=cfg["actnorm"], split_prior=cfg["split_prior"]).cuda()
            step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), grad_clip_norm=1.0, autocast=True, graph=graph,
                             graph_warmup=2, batch_cond_prep=batched)
            seqs[graph, batched] = [float(step(x)) for _ in range(6)]
            couplings = [m for m in model.modules() if isinstance(m, Coupling)]
            assert couplings and all(m._prepared_wt is None for m in couplings)
            if batched:
                assert step.cond_prep.table is not None and step.cond_prep.table.n == len(couplings)
                if graph:
                    assert step._captured is not None
    assert seqs[False, True] == seqs[False, False] and seqs[True, True] == seqs[True, False]
    assert seqs[True, True][-1] < seqs[True, True][0]


def test_clear_grads_keeps_the_bucket_and_changes_nothing(fixture):
    """TrainStep(clear_grads=True) masks the inverse-flow layers' gradients in place (they are views of the flat bucket): a run
    with it and a run without it take the same steps -- the library's dW is masked already -- eagerly and as a captured
    graph, and every gradient still lives in the bucket afterwards (a replaced .grad would neither be zeroed nor
    all-reduced, and a captured backward would keep accumulating into the warm-up tensor).  With graph=True the learning
    rate is a device tensor: set_lr takes effect in the next replay."""
    from inf.train.step import TrainStep
    x = torch.from_numpy(fixture["x"]).float().cuda()
    finals = {}
    for graph in (False, True):
        for clear in (False, True):
            torch.manual_seed(0)
            model = build(fixture)
            step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), grad_clip_norm=1.0, clear_grads=clear, graph=graph,
                             graph_warmup=2)
            losses = [float(step(x)) for _ in range(6)]
            lo, hi = step.bucket.flat.data_ptr(), step.bucket.flat.data_ptr() + step.bucket.flat.numel() * 4
            assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
            finals[(graph, clear)] = (losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]))
        la, pa = finals[(graph, False)]
        lb, pb = finals[(graph, True)]
        assert la == lb and torch.equal(pa, pb), graph
    # the captured step reads the learning rate from the device
    torch.manual_seed(0)
    model = build(fixture)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), graph=True, graph_warmup=1)
    for _ in range(3):
        step(x)
    assert step._captured is not None and torch.is_tensor(step.optimizer.param_groups[0]["lr"])
    step.set_lr(0.0)
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    step(x)
    assert torch.equal(before, torch.cat([p.detach().reshape(-1) for p in model.parameters()]))
    # ... also when a scheduler ASSIGNS a number to group['lr'] (torch's schedulers do): the value goes into the device tensor
    lr_t = step.optimizer.param_groups[0]["lr"]
    step.optimizer.param_groups[0]["lr"] = 1e-3
    step(x)
    assert step.optimizer.param_groups[0]["lr"] is lr_t and abs(float(lr_t) - 1e-3) < 1e-9
    assert not torch.equal(before, torch.cat([p.detach().reshape(-1) for p in model.parameters()]))


COLLECTIVE_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))   # RCCL, one rank: the collective is real, its value a no-op
from test_hip_train import build
from inf.train.step import TrainStep
d = np.load(%r); fixture = {k: d[k] for k in d.files}
x = torch.from_numpy(fixture["x"]).float().cuda()
seqs = []
for graph, force in ((True, False), (True, True)):  # (both with the capturable optimizer: the same arithmetic)
    torch.manual_seed(0)
    model = build(fixture)
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), grad_clip_norm=1.0, graph=graph, graph_warmup=2,
                     force_collective=force)
    seqs.append([float(step(x)) for _ in range(8)])
    assert step._captured is not None and step._split == force and (step._captured2 is not None) == force
assert seqs[0] == seqs[1], seqs
dist.destroy_process_group()
print("captured step with the collective: ok")
'''


def test_captured_step_with_the_bucket_all_reduce_inside(tmp_path):
    """TrainStep(graph=True) in a process group (RCCL, backend "nccl", here a group of one rank: the collective kernel is
    issued, its result is the identity): the step replays as TWO captured graphs around the eagerly issued all-reduce of the
    flat gradient bucket -- a captured RCCL collective makes ProcessGroupNCCL's watchdog query an event recorded in a
    capturing stream, which ends the process on this stack (inf/train/step.py) -- and gives, bit for bit, the loss sequence
    of the one-graph step without the collective.  Its own process: the process group must not leak into the suite."""
    import subprocess
    import sys
    from conftest import PKG, ROOT
    script = tmp_path / "w.py"
    script.write_text(COLLECTIVE_WORKER % (PKG, ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN, "trainstep_glow_b6_8x8_L2K2.npz")))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "captured step with the collective: ok" in r.stdout, r.stdout[-3000:]


def test_flat_adam_matches_the_multi_tensor_optimizer(fixture):
    """TrainStep(graph=True) runs Adam as one pass over a flat parameter buffer (ifl_adam_flat_f32): parameters re-homed as
    views, state entered in the optimizer.  Against the same steps on torch's fused multi-tensor Adam: the same losses and the
    same parameters to rounding (1e-6 of each tensor's scale after 6 steps); the parameters are views of one buffer, the
    shared-storage pair of every Conv2dZero keeps its storage, and the optimizer's state_dict has an entry per parameter."""
    from inf.train.step import TrainStep
    x = torch.from_numpy(fixture["x"]).float().cuda()
    runs = []
    for flat in (True, False):
        torch.manual_seed(0)
        model = build(fixture)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
        step = TrainStep(model, opt, grad_clip_norm=1.0, graph=True, graph_warmup=2, flat_optimizer=flat)
        losses = [float(step(x)) for _ in range(6)]
        runs.append((losses, [p.detach().clone() for p in model.parameters()], step, opt, model))
    (l1, p1, step1, opt1, model1), (l0, p0, _, _, _) = runs
    assert step1._flat is not None
    assert np.allclose(l1, l0, rtol=1e-6, atol=0)
    for a, b in zip(p1, p0):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
    lo, hi = step1._flat["p"].data_ptr(), step1._flat["p"].data_ptr() + step1._flat["p"].numel() * 4
    inside = [lo <= p.data_ptr() < hi for p in model1.parameters()]
    assert sum(inside) >= len(inside) - 2 * sum(1 for m in model1.modules() if type(m).__name__ == "Conv2dZero") and any(inside)
    sd = opt1.state_dict()["state"]
    assert len(sd) == sum(inside) and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd.values())

