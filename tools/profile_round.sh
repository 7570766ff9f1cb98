#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel stats, and the two HBM counter passes
# (separate --pmc runs, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-train-step > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-train-step > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-train-step > $OUT/pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/pmc_sq -o sq -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-train-step > $OUT/pmc_sq.log 2>&1
echo "sq done"
rocprofv3 --kernel-trace --stats -d $OUT/stats_wide -o stats -- python3 $ROOT/tools/prof_wide.py 16,256,8,8,3 > $OUT/stats_wide.log 2>&1
echo "wide done"
rocprofv3 --kernel-trace --stats -d $OUT/stats_small -o stats -- python3 $ROOT/tools/prof_small.py 64,1,28,28,3 100,4,14,14,2 100,8,7,7,2 100,12,16,16,2 100,24,8,8,2 100,48,4,4,2 > $OUT/stats_small.log 2>&1
echo "small done"
find $OUT -name "*.csv" | head -20
