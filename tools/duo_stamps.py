"""Timeline of the duo scan (scan_duo.hip) from its in-kernel stamps (development aid).

    python inverse-flow_amd/build.py --stamps      # lib/libinvflow_hip_stamps.so, here or on the GPU box
    python tools/duo_stamps.py [B]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
os.environ["IFL_STAMPS"] = str(buf.data_ptr())
import invflow_hip as H
H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", "libinvflow_hip_stamps.so")
from bench import ref_init_weight, B, C, HH, WW
if len(sys.argv) > 1:
    B = int(sys.argv[1])
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
for _ in range(5):
    H.inverse(x, w, out=z)
torch.cuda.synchronize()
s = buf.cpu().tolist()
t0 = s[0]
for part, name in ((0, "upper / single"), (1, "lower")):
    o = s[32 * part:32 * part + 32]
    if not any(o):
        continue
    steps = o[4]
    print("%-15s entry %+7.2f us  sweep start %+7.2f us  sweep end %+7.2f us   %d steps, %d cycles per step, clock %.2f GHz"
          % (name, (o[0] - t0) / 100.0, (o[1] - t0) / 100.0, (o[2] - t0) / 100.0, steps, o[3] // max(steps, 1),
             o[3] / max((o[2] - o[1]) * 10.0, 1.0)))
    print("   chain wave 0: %d cycles per step at the barrier" % (o[5] // max(steps, 1)))
    print("   chain wave 0 cycles per step:", {n_: o[24 + k] // max(steps, 1) for k, n_ in enumerate(
        ["barrier", "leading products + requests", "critical products", "-", "trailing products + epilogue + drained ring write"]) if n_ != "-"})
    print("   helper 0: slow polls %d (spins %d), gate wait %.2f us" % (o[8], o[9], o[10] / 100.0))
    names = ["barrier", "products 1", "z, requests", "memory requests", "hand-off in, seeds", "products 2", "final wait"]
    print("   helper 0 cycles per step:", {n: o[16 + k] // max(steps, 1) for k, n in enumerate(names)})

# per-step timeline of the chain wave 0 of image 0's workgroups: cycles between consecutive barriers, and of those the wait
for part, name in ((0, "upper / single"), (1, "lower")):
    o = s[128 + 128 * part:256 + 128 * part]
    if not any(o):
        continue
    t = [o[2 * k] for k in range(64) if o[2 * k]]
    w = [o[2 * k + 1] for k in range(64) if o[2 * k]]
    print(name, "steps between barriers (cycles; in brackets: of which waiting at the barrier):")
    print("  " + " ".join("%d[%d]" % (t[k + 1] - t[k], w[k + 1]) for k in range(len(t) - 1)))
# k_foldpack (workgroup 0): cycles of its phases (IFL_FSTAMP in scan_mfma.hip)
f = s[160:164]
if any(f):
    print("k_foldpack workgroup 0 cycles: loads %d, diagonal blocks %d, block solve %d, pack %d" % tuple(f))
