"""Time the bench step (inverse + fused backward) of the north-star workload on what-if builds (tools/exp_scan.sh).

    python tools/time_step.py [name ...]      # lib/libinvflow_hip_<name>.so; no name: the product library"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
    import torch
    import invflow_hip as H
    name = sys.argv[2]
    if name != "product":
        H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", "libinvflow_hip_%s.so" % name)
    from bench import ref_init_weight, B, C, HH, WW
    gen = torch.Generator().manual_seed(0)
    w = ref_init_weight(gen).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
    z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
    carry = H.new_carry(w)
    def step():
        H.inverse(x, w, "TL", 0, out=z, carry=carry)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1000)
    print("%-24s %7.1f us per step (inverse + backward)" % (name, best), flush=True)
else:
    for name in (sys.argv[1:] or ["product"]):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name])
