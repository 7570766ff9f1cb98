#!/bin/bash
# SQ counters of a short bench run (one --pmc pass, kernel trace only): MFMA busy cycles, wave cycles, wait buckets,
# LDS bank conflicts.  tools/pmc_sq.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_sq_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq -o sq -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/sq.log 2>&1
tail -2 $OUT/sq.log
python3 $ROOT/tools/rocpd_summary.py $OUT/sq/sq_results.db | cut -c1-170
