#!/bin/bash
# Build what-if variants of the wide-layer kernels (csrc/wide.hip) next to the product library:
#   tools/exp_wide.sh name:"-DIFL_WIDE_SCOPE=__HIP_MEMORY_SCOPE_AGENT" ...  ->  inverse-flow_amd/lib/libinvflow_hip_<name>.so
# (time them with tools/prof_wide.py --lib libinvflow_hip_<name>.so)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/inverse-flow_amd
python build.py > /dev/null
for a in "$@"; do
  e=${a%%:*}; f=${a#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -Wno-inline-asm $f -c csrc/wide.hip -o build/wide_x$e.o 2>build/wide_x$e.log || { grep -m3 error build/wide_x$e.log; rm -f build/wide_x$e.o; } &
done
wait
for a in "$@"; do
  e=${a%%:*}
  objs=$(ls build/*.o | grep -v "build/wide" | grep -v scan_duo_x)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libinvflow_hip_$e.so $objs build/wide_x$e.o && echo lib/libinvflow_hip_$e.so
done
