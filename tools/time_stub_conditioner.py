"""upper bound of what a fused conditioner could buy: the configs[2]/[3] steps with every coupling's conditioner replaced by a
three-node stand-in (graph replay)"""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.train.step import TrainStep
from inf.layers.coupling import Coupling
if len(sys.argv) > 1 and sys.argv[1] == "stub":
    def stub(self, x, context):
        c3 = self.net[4]
        return (x[:, :1] * 0.0).expand(-1, self.n_channels, -1, -1) + c3.bias[None, :, None, None]
    Coupling._conditioner = stub
for which in ("mnist", "cifar"):
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
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True)
    for _ in range(6):
        loss = step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        loss = step(x)
    torch.cuda.synchronize()
    print("%s %s: %.2f ms per step" % (which, sys.argv[1:] or "full", (time.perf_counter() - t0) / 10 * 1e3), flush=True)
