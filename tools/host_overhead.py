"""Host-side cost of the Python wrappers (launch-bound regime): us per call without waiting for the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
x = torch.randn(4, 12, 16, 16, device="cuda"); g = torch.randn_like(x); gl = torch.randn(4, device="cuda")
t = torch.randn(12, device="cuda"); ls = torch.randn(12, device="cuda")
w = (torch.eye(12).view(12, 12, 1, 1) * torch.tensor([[0., 0.], [0., 1.]]) + 0.01 * torch.randn(12, 12, 2, 2)).cuda()
z = H.inverse(x, w)


def bench(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-28s host %6.1f us/call   (device drained after %6.1f us/call)" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))


bench("torch.empty_like", lambda: torch.empty_like(x))
bench("x + 1 (one eager kernel)", lambda: x + 1)
bench("H.slr", lambda: H.slr(x, 0.3))
bench("H.slr_backward", lambda: H.slr_backward(g, gl, x, 0.3))
bench("H.actnorm", lambda: H.actnorm(x, t, ls))
bench("H.actnorm_backward", lambda: H.actnorm_backward(g, gl, x, t, ls))
bench("H.inverse (2x2, C=12)", lambda: H.inverse(x, w))
bench("H.backward", lambda: H.backward(g, z, w))
