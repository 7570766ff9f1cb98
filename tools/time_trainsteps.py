"""configs[2] / configs[3] training steps (graph replay) with and without MIOpen's find mode (torch.backends.cudnn.benchmark)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import importlib
from inf.train.step import TrainStep
bm = len(sys.argv) > 1 and sys.argv[1] == "1"
fused = os.environ.get("FUSED", "1") == "1"
torch.backends.cudnn.benchmark = bm
from inf.layers.coupling import Coupling
Coupling.channels_last = os.environ.get("NHWC", "1") == "1"
for which in (sys.argv[2:] or ["mnist", "cifar"]):
    mod = importlib.import_module("inf.experiments.if_glow_" + which)
    cfg = mod.DEFAULT_CONFIG
    torch.manual_seed(4)
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
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True, conv_search=bm, fused_optimizer=fused,
                     batch_cond_prep=os.environ.get("BATCH_PREP", "1") == "1")
    for _ in range(6):
        loss = step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loss = step(x)
    torch.cuda.synchronize()
    print("%s nhwc=%s cudnn.benchmark=%s fused=%s: %.2f ms per step, loss %.4f" % (which, Coupling.channels_last, bm, fused, (time.perf_counter() - t0) / 10 * 1e3, float(loss)), flush=True)
