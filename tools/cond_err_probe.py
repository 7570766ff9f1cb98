import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/inverse-flow_amd"); sys.path.insert(0, "/root/repo")
import torch
import test_hip_conditioner as T
for (B, C, H, W, width) in T.SHAPES[:5]:
    layer = T._coupling(C, width, seed=B + C)
    torch.manual_seed(7)
    x = torch.randn(B, C, H, W, device="cuda")
    gy, gld = torch.randn(B, C, H, W, device="cuda"), torch.randn(B, device="cuda")
    y1, ld1, gx1, gp1 = T._run(layer, x, gy, gld, fused=True)
    y0, ld0, gx0, gp0 = T._run(layer, x, gy, gld, fused=False)
    # fp64 reference of the module tree
    l64 = __import__("copy").deepcopy(layer).double()
    from inf.layers.coupling import Coupling
    Coupling.fused = False
    xin = x.double().clone().requires_grad_()
    x1, x2, log_s, t = l64.get_xs_logs_t(xin)
    y = torch.cat([x1, torch.addcmul(t, x2, log_s.exp())], dim=1); ld = log_s.sum(dim=(1, 2, 3))
    (y * gy.double()).sum().add((ld * gld.double()).sum()).backward()
    Coupling.fused = True
    c1, c2, c3 = l64.net[0], l64.net[2], l64.net[4]
    ref = [p.grad for p in (c1.weight, c2.weight, c3.weight, c3.bias, c3.logs)]
    def e(a, b): return float((a.double() - b).abs().max() / b.abs().max())
    print((B, C, H, W, width), "gx fused %.1e torch %.1e |" % (e(gx1, xin.grad), e(gx0, xin.grad)),
          " ".join("%s fused %.1e torch %.1e" % (n, e(a, r), e(b, r)) for n, a, b, r in zip(("dW1", "dW2", "dW3", "db", "dlogs"), gp1, gp0, ref)), flush=True)
