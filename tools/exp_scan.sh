#!/bin/bash
# Build what-if variants of the duo scan (scan_duo.hip) next to the product library:
#   tools/exp_scan.sh name:"-DIFL_EXP=3" name2:"-DIFL_PRIO_CHAIN=2 ..."  ->  inverse-flow_amd/lib/libinvflow_hip_<name>.so
# (time them with tools/time_scan.py)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/inverse-flow_amd
python build.py > /dev/null
for a in "$@"; do
  e=${a%%:*}; f=${a#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -fno-slp-vectorize $f -c csrc/scan_duo.hip -o build/scan_duo_x$e.o 2>build/scan_duo_x$e.log || { grep -m3 error build/scan_duo_x$e.log; rm -f build/scan_duo_x$e.o; } &
done
wait
for a in "$@"; do
  e=${a%%:*}
  objs=$(ls build/*.o | grep -v scan_duo | grep -v "build/wide_x")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libinvflow_hip_$e.so $objs build/scan_duo_x$e.o && echo lib/libinvflow_hip_$e.so
done
