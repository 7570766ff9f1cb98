#!/bin/bash
# A/B two prebuilt libraries on the model training steps, same box: tools/ab_steps.sh a.so b.so  (alternating, 3 rounds;
# IMAGENET32=1 adds the configs[4] model at 13 images per rank)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for L in "$@"; do
  cp $ROOT/$L $ROOT/inverse-flow_amd/lib/libinvflow_hip.so
  echo "$L: $(timeout -k 10 100 python3 $ROOT/tools/time_trainsteps.py 2>/dev/null | grep 'ms per step' | sed 's/nhwc.*fused=True//' | tr '\n' ' ')"
  if [ -n "$IMAGENET32" ]; then echo "$L: $(timeout -k 10 200 python3 $ROOT/tools/time_imagenet32_step.py 2>/dev/null | grep -i 'ms' | tr '\n' ' ')"; fi
done; done
