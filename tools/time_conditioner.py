"""fused coupling conditioner: forward + backward of one Coupling layer at the models' shapes, many times (run under
rocprofv3 --kernel-trace and read tools/kernel_breakdown.py for the per-kernel durations)
usage: time_conditioner.py [--autocast] B,C,H,W,width [...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.layers.coupling import Coupling
args = sys.argv[1:]
ac = "--autocast" in args
args = [a for a in args if a != "--autocast"]
for spec in args or ["32,24,8,8,128"]:
    B, C, H, W, width = (int(v) for v in spec.split(","))
    torch.manual_seed(0)
    layer = Coupling((C, H, W), width=width).cuda()
    with torch.no_grad():
        layer.net[4].weight.normal_(0, 0.05)
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
            y, ld = layer(x)
        (y.sum() + ld.sum()).backward()
    for _ in range(3):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 30
    for _ in range(N):
        run()
    torch.cuda.synchronize()
    print("%s autocast=%s: %.1f us per forward+backward (eager, host-bound)" % (spec, ac, (time.perf_counter() - t0) / N * 1e6), flush=True)
