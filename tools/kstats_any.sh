#!/bin/bash
# kernel-trace statistics of any script: tools/kstats_any.sh <tag> <script.py> [args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/"$@" > $OUT/stats.log 2>&1
tail -3 $OUT/stats.log
python3 $ROOT/tools/rocpd_summary.py $OUT/stats/stats_results.db | cut -c1-160 | head -30
