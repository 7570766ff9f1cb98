#!/bin/bash
# A/B two prebuilt libraries on the same box: tools/ab.sh a.so b.so  (alternating, 3 rounds)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for L in "$@"; do
  cp $ROOT/$L $ROOT/inverse-flow_amd/lib/libinvflow_hip.so
  echo -n "$L: "; timeout -k 10 100 python3 $ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']),d['ms_per_step'],d['roofline']['per_kernel_us'])"
done; done
