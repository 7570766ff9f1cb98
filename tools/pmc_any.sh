#!/bin/bash
# SQ counters of any script: tools/pmc_any.sh <tag> <script.py>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/sq -o sq -- python3 $ROOT/"$@" > $OUT/sq.log 2>&1
python3 $ROOT/tools/rocpd_summary.py $OUT/sq/sq_results.db | grep -E "conv_mfma|wgrad_mfma" | cut -c1-400
