#!/bin/bash
# kernel-trace statistics of a short bench run: tools/kstats.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ks_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > $OUT/stats.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/stats/stats_results.db | cut -c1-160 | head -20
