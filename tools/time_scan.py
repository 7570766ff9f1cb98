"""Time the inverse scan alone at the north-star shape, for one or more builds of the library:
    python tools/time_scan.py [libinvflow_hip.so libinvflow_hip_exp3.so ...]      (names under inverse-flow_amd/lib)
Each build runs in its own process (the library is loaded once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
    import torch
    import invflow_hip as H
    H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", sys.argv[2])
    from bench import ref_init_weight, B, C, HH, WW
    w = ref_init_weight(torch.Generator().manual_seed(0)).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
    for _ in range(20):
        H.inverse(x, w, out=z)
    res = []
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            H.inverse(x, w, out=z)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 10.0)
    res.sort()
    print("%-32s inverse (fold + scan): median %.1f us, min %.1f us per call" % (sys.argv[2], res[2], res[0]))
else:
    libs = sys.argv[1:] or ["libinvflow_hip.so"]
    for rnd in range(2):
        for lib in libs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib])
