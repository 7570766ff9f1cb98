"""Time the inverse (fold + scan) of the bench workload on what-if builds of the library (tools/exp_scan.sh).

    python tools/time_scan.py [name ...]      # lib/libinvflow_hip_<name>.so; no name: the product library
Each library runs in its own process (a process loads one)."""
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
    x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
    for _ in range(20):
        H.inverse(x, w, out=z)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            H.inverse(x, w, out=z)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1000)
    print("%-24s %7.1f us per inverse (fold + scan)" % (name, best), flush=True)
else:
    for name in (sys.argv[1:] or ["product"]):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name])
