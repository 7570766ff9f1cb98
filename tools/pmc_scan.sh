#!/bin/bash
# HBM-side counters of the scan only (two separate --pmc passes); prints the per-launch means.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f -o f -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/w -o w -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/w.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/f/f_results.db $OUT/w/w_results.db | grep -E "scan_mfma" | cut -c1-150
