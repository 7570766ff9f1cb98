"""Soak of the split scan: many back-to-back launches; results stay bit-identical and no batch of launches shows the
signature of a failed hand-off (a timed-out poll costs milliseconds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW
w = ref_init_weight(torch.Generator().manual_seed(0)).cuda()
x = torch.randn(B, C, HH, WW, device="cuda")
ref = H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE)
z = torch.empty_like(x)
N, CH = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 100
times = []
for c in range(N // CH):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(CH):
        H.inverse(x, w, out=z)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / CH * 1e6)
    if c % 20 == 0:
        assert torch.equal(z, ref), c
assert torch.equal(z, ref)
times.sort()
print("%d launches: per-launch time of a %d-launch batch: median %.1f us, max %.1f us" % (N, CH, times[len(times) // 2], times[-1]))
