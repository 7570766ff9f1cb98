"""Soak of the wide layers' team scan: many back-to-back launches at two shapes in turn; results stay identical to the first
launch's, the launch-per-diagonal route agrees within the tolerance, and no launch is voided."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cases = []
for (B, C, HH, WW, K) in [(16, 256, 8, 8, 3), (40, 256, 8, 8, 3), (12, 128, 16, 16, 3)]:
    w = torch.zeros(C, C, K, K); w[:, :, -1, -1] = torch.eye(C)
    w = (w + 0.01 * torch.randn(C, C, K, K)).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda")
    ref = H.inverse(x, w)
    slow = H.inverse(x, w, "TL", H.FLAG_WHOLE_IMAGE)
    assert float((ref - slow).norm() / slow.norm()) < 1e-5
    cases.append((x, w, ref, torch.empty_like(x)))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    x, w, ref, z = cases[i % len(cases)]
    H.inverse(x, w, out=z)
    if i % 500 == 499:
        torch.cuda.synchronize()
        for (x, w, ref, z) in cases:
            assert torch.equal(z, ref), i
torch.cuda.synchronize()
print("%d team-scan launches over %d shapes in %.2f s, voided %d" % (N, len(cases), time.perf_counter() - t0, H.scan_voided(cases[0][0].device)))
