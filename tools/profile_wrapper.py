"""cProfile of a wrapper call (where do the ~14 us of host time per call go?)."""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
x = torch.randn(4, 12, 16, 16, device="cuda")
for _ in range(100):
    H.slr(x, 0.3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    H.slr(x, 0.3)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
