#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run into profiles/ (tracked): tools/publish_profiles.sh <tag>
set -e
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=gpurun_out/prof_$TAG
cd $ROOT
cp $SRC/bench.json profiles/r01_final_bench.json
python tools/rocpd_summary.py $SRC/stats/stats_results.db > profiles/r01_final_rocprofv3_kernel_stats.txt
python tools/rocpd_summary.py $SRC/pmc_fetch/fetch_results.db $SRC/pmc_write/write_results.db > profiles/r01_final_rocprofv3_pmc_hbm.txt
if [ -f $SRC/pmc_sq/sq_results.db ]; then
  python tools/rocpd_summary.py $SRC/pmc_sq/sq_results.db > profiles/r01_final_rocprofv3_pmc_sq.txt
fi
python - "$SRC" <<'PY'
import json, re, sys
src = sys.argv[1]
txt = open("profiles/r01_final_rocprofv3_pmc_hbm.txt").read()
def mean(counter):
    m = re.search(r"k_scan_(?:split|mfma)\S*\s+%s=([0-9.]+) \(n=(\d+)\)" % counter, txt)
    return float(m.group(1)), int(m.group(2))
f, n = mean("FETCH_SIZE"); w, _ = mean("WRITE_SIZE")
name = re.search(r"(k_scan_(?:split|mfma))", txt).group(1)
json.dump({
    "kernel": name + "<64,3,3> (B=128: two workgroups per image)" if name == "k_scan_split" else name,
    "workload": "B=128, C=64, 32x32, K=3 (bench.py)",
    "source": "profiles/r01_final_rocprofv3_pmc_hbm.txt: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), mean of %d launches" % n,
    "fetch_size_kib": f, "write_size_kib": w, "fetch_correction": 2.0,
    "correction_note": "MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads; WRITE_SIZE is exact for 16-B-per-lane stores",
}, open("profiles/r01_final_scan_hbm_counters.json", "w"), indent=2)
PY
python - <<'PY'
import json, os, re
p = "profiles/r01_final_rocprofv3_pmc_sq.txt"
if os.path.exists(p):
    txt = open(p).read()
    m = re.search(r"k_scan_(?:split|mfma)\S*\s+.*?SQ_BUSY_CYCLES=([0-9.]+).*?SQ_VALU_MFMA_BUSY_CYCLES=([0-9.]+)", txt)
    if m:
        busy, mfma = float(m.group(1)), float(m.group(2))
        json.dump({"kernel": "scan", "source": p + " (rocprofv3 --pmc, SQ pass; counters per shader engine = 32 SIMDs)",
                   "sq_busy_cycles": busy, "sq_valu_mfma_busy_cycles": mfma, "mfma_busy_frac": mfma / (32.0 * busy)},
                  open("profiles/r01_final_sq_counters.json", "w"), indent=2)
PY
ls -la profiles/
