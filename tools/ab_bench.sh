#!/bin/bash
# A/B two prebuilt libraries on a bench.py workload, same box: tools/ab_bench.sh "<bench args>" a.so b.so  (alternating, 2 rounds)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for i in 1 2; do for L in "$@"; do
  cp $ROOT/$L $ROOT/inverse-flow_amd/lib/libinvflow_hip.so
  echo -n "$L: "; timeout -k 10 200 python3 $ROOT/bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']),round(d['ms_per_step'],3))"
done; done
