"""torch.profiler over eager training steps of a configs[] model: which aten ops (and how many of them) a step consists of"""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from inf.train.step import TrainStep
which = sys.argv[1] if len(sys.argv) > 1 else "cifar"
torch.backends.cudnn.benchmark = True
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
step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=cfg["lr"]), grad_clip_norm=1.0, autocast=True, graph=True, graph_warmup=10**9)
for _ in range(4):
    step(x)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step(x)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("%s: device time per step %.2f ms" % (which, tot / N / 1e3))
for e in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    if e.self_device_time_total <= 0:
        break
    print("%6.2f%% %6.1f calls/step %8.1f us/step  %s" % (100 * e.self_device_time_total / tot, e.count / N, e.self_device_time_total / N, e.key[:110]))
print("parameters: %d tensors" % len(list(model.parameters())))
