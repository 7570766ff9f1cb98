#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/ (tracked) under the round's
name, and point profiles/LATEST.json at them: bench.py takes `roofline.traffic` and `mfma_busy_frac` from there, and only
when the kernel the counters were taken on is the one it just timed.

    python tools/publish_profiles.py <tag> <name>        e.g.  r02a r02
"""
import json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out = os.path.join(ROOT, "profiles")


def summary(*dbs):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py")] + list(dbs), check=True,
                          stdout=subprocess.PIPE, text=True).stdout


shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, name + "_bench.json"))
stats = summary(os.path.join(src, "stats", "stats_results.db"))
open(os.path.join(out, name + "_rocprofv3_kernel_stats.txt"), "w").write(stats)
hbm = summary(os.path.join(src, "pmc_fetch", "fetch_results.db"), os.path.join(src, "pmc_write", "write_results.db"))
open(os.path.join(out, name + "_rocprofv3_pmc_hbm.txt"), "w").write(hbm)
sq = ""
if os.path.exists(os.path.join(src, "pmc_sq", "sq_results.db")):
    sq = summary(os.path.join(src, "pmc_sq", "sq_results.db"))
    open(os.path.join(out, name + "_rocprofv3_pmc_sq.txt"), "w").write(sq)

if os.path.exists(os.path.join(src, "stats_wide", "stats_results.db")):
    open(os.path.join(out, name + "_rocprofv3_kernel_stats_wide_layer.txt"), "w").write(
        summary(os.path.join(src, "stats_wide", "stats_results.db")))

if os.path.exists(os.path.join(src, "stats_small", "stats_results.db")):
    open(os.path.join(out, name + "_rocprofv3_kernel_stats_small_layers.txt"), "w").write(
        summary(os.path.join(src, "stats_small", "stats_results.db")))

SCAN = r"(\S*k_scan_(?:duo|mfma)\S*)"
m = re.search(SCAN + r"\s+(\d+)\s+([0-9.]+)", stats)
kernel, calls, avg_us = m.group(1), int(m.group(2)), float(m.group(3))


def counter(txt, cname):
    mm = re.search(SCAN + r".*?%s=([0-9.]+) \(n=(\d+)\)" % cname, txt)
    return float(mm.group(2)), int(mm.group(3))


f, n = counter(hbm, "FETCH_SIZE")
w, _ = counter(hbm, "WRITE_SIZE")
latest = {
    "round": name, "kernel": kernel, "workload": "B=128, C=64, 32x32, K=3 (bench.py)", "scan_avg_us": avg_us, "scan_calls": calls,
    "fetch_size_kib": f, "write_size_kib": w, "fetch_correction": 2.0,
    "correction_note": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads; WRITE_SIZE is exact for 16-B-per-lane stores",
    "source": "profiles/%s_rocprofv3_pmc_hbm.txt (two separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE; mean of %d launches), "
              "profiles/%s_rocprofv3_kernel_stats.txt" % (name, n, name),
}
mm = re.search(SCAN + r".*?SQ_BUSY_CYCLES=([0-9.]+).*?SQ_VALU_MFMA_BUSY_CYCLES=([0-9.]+)", sq)
if mm:
    busy, mfma = float(mm.group(2)), float(mm.group(3))
    latest.update(sq_busy_cycles=busy, sq_valu_mfma_busy_cycles=mfma, mfma_busy_frac=mfma / (32.0 * busy),
                  sq_source="profiles/%s_rocprofv3_pmc_sq.txt (counters per shader engine = 32 SIMDs)" % name)
json.dump(latest, open(os.path.join(out, "LATEST.json"), "w"), indent=2)
print(json.dumps(latest, indent=2))
