"""Per-section cycle counts of the MFMA scan's step (development aid).

Build the library with the stamps compiled in, run on the GPU box, then rebuild without:
    HIPCC_EXTRA=-DIFL_STAMPS python inverse-flow_amd/build.py --force && python tools/stamps.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(176, dtype=torch.int64, device="cuda")
os.environ["IFL_STAMPS"] = str(buf.data_ptr())
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
for _ in range(3):
    H.inverse(x, w, out=z)
torch.cuda.synchronize()
both = buf.cpu()
for part in range(2):
  full = both[80 * part:80 * part + 80]
  if int(full.abs().sum()) == 0:
      continue
  print("== part", part, "(whole image)" if int(both[80:].abs().sum()) == 0 else "(split scan)")
  print("   begin", int(full[78]), "end", int(full[79]), "span", int(full[79] - full[78]), "memtime ticks; halo slow paths", int(full[76]), "polls", int(full[77]))
  rt = [int(v) for v in full[72:76]]
  print("   realtime (10 ns ticks, relative to part 0's start): begin", rt[0] - int(both[72]), "sweep start", rt[1] - int(both[72]), "end", rt[2] - int(both[72]), "upper half: step 18 done / lower half: kernel entry", rt[3] - int(both[72]) if rt[3] else None)
  t = full[:64].view(8, 8)
  names = ["dma", "wait+bar", "reads+lead", "crit", "epilogue", "trail", "stores", "loop"]
  for wv in range(8):
      r = t[wv].tolist()
      if sum(r) == 0:
          continue
      print("wave", wv, {names[k]: r[k] for k in range(8)}, "total", sum(r))
  for k, name in enumerate(["no tile", "tile 0", "tile 1", "both tiles"]):
      c = int(full[68 + k]); tot = int(full[64 + k])
      if c:
          print("steps with", name, ":", c, "steps,", tot // c, "cycles each")
print("fold (workgroup 0):", dict(zip(["loads", "diagonal blocks", "block solve", "pack"], both[160:164].tolist())))
