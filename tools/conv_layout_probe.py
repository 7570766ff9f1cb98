"""MIOpen under torch: NCHW vs channels_last bf16 convolutions of the coupling conditioners' shapes (collected from the
configs[2]/[3] models) -- device time and number of kernels for forward + backward, and whether a naive_conv kernel was picked"""
import os, sys, importlib, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
shapes = {}
orig = F.conv2d
def spy(x, w, *a, **k):
    if x.dim() == 4 and w.shape[1] != w.shape[0] or w.shape[-1] in (1, 3):
        key = (x.shape[0], w.shape[1], w.shape[0], w.shape[-1], x.shape[-1])
        shapes[key] = shapes.get(key, 0) + 1
    return orig(x, w, *a, **k)
F.conv2d = spy
torch.nn.functional.conv2d = spy
for which in ("mnist", "cifar"):
    mod = importlib.import_module("inf.experiments.if_glow_" + which)
    cfg = mod.DEFAULT_CONFIG
    if which == "cifar":
        model = mod.create_model(inv_flow=cfg["inv_flow"], inv_conv=cfg["inv_conv"], inv_conv_no_pad=cfg["inv_conv_no_pad"],
                                 if_kernel_size=cfg["if_kernel_size"], num_blocks=cfg["num_blocks"], block_size=cfg["block_size"],
                                 coupling_width=cfg["coupling_width"], activation=cfg["activation"], actnorm=cfg["actnorm"],
                                 split_prior=cfg["split_prior"]).cuda()
        x = torch.randint(0, 256, (32, 3, 32, 32), device="cuda").float()
    else:
        model = mod.create_model(num_blocks=cfg["num_blocks"], block_size=cfg["block_size"], coupling_width=cfg["coupling_width"],
                                 n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"]).cuda()
        x = torch.randint(0, 256, (cfg["batch_size"], 1, 28, 28), device="cuda").float()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        model(x)
    del model
F.conv2d = orig
torch.nn.functional.conv2d = orig
for (B, ci, co, k, hw), cnt in sorted(shapes.items()):
    res = []
    for cl in (False, True):
        x = torch.randn(B, ci, hw, hw, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(co, ci, k, k, device="cuda", dtype=torch.bfloat16)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(); w.requires_grad_()
        y = F.conv2d(x, w, padding=k // 2)
        g = torch.ones_like(y)
        for _ in range(3):
            y = F.conv2d(x, w, padding=k // 2)
            y.backward(g)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                y = F.conv2d(x, w, padding=k // 2)
                y.backward(g)
            torch.cuda.synchronize()
        ev = [e for e in prof.key_averages() if e.self_device_time_total > 0]
        tot = sum(e.self_device_time_total for e in ev) / 5
        n = sum(e.count for e in ev) / 5
        naive = sum(e.self_device_time_total for e in ev if "naive" in e.key) / 5
        res.append("%s %6.1f us %2.0f kernels (naive %5.1f us)" % ("NHWC" if cl else "NCHW", tot, n, naive))
    print("x%d B%d %3d->%3d k%d %2dx%2d: %s | %s" % (cnt, B, ci, co, k, hw, hw, res[0], res[1]), flush=True)
