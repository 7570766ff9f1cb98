#!/bin/bash
# Kernel census of the three model training steps (graph replay) on the GPU box: a rocprofv3 kernel trace per model, the
# steady tail of each through tools/kernel_breakdown.py -> gpurun_out/prof_models/model_step_kernels.txt; and the
# configs[3] / configs[4] full-batch bench lines.  (Copy the results into profiles/ afterwards.)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_models
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$OUT/model_step_kernels.txt
: > $R
for M in cifar mnist; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$M -o t -- python3 $ROOT/tools/time_trainsteps.py 0 $M > $OUT/$M.log 2>&1
  echo "== configs[$M] training step, graph replay: rocprofv3 --kernel-trace, the last 60 ms of the run (tools/kernel_breakdown.py)" >> $R
  python3 $ROOT/tools/kernel_breakdown.py $OUT/$M 40 60 >> $R
  echo "$M done"
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/imagenet32 -o t -- python3 $ROOT/tools/time_imagenet32_step.py --graph-only > $OUT/imagenet32.log 2>&1
echo "== configs[4] model (if_multiGPU_imagenet32, 13 images per rank) training step as configured, graph replay: the last 120 ms of the run" >> $R
python3 $ROOT/tools/kernel_breakdown.py $OUT/imagenet32 40 120 >> $R
echo "imagenet32 done"
python3 $ROOT/bench.py --workload cifar_step > $OUT/bench_cifar_step.json 2> $OUT/bench_cifar_step.err
python3 $ROOT/bench.py --workload imagenet32_step > $OUT/bench_imagenet32_step.json 2> $OUT/bench_imagenet32_step.err
echo "bench lines done"
rm -rf $OUT/cifar $OUT/mnist $OUT/imagenet32
