#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ks_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/prof_small.py
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/tools/prof_small.py > $OUT/stats.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/stats/stats_results.db | cut -c1-160 | head -24
