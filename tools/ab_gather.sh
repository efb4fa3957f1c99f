#!/bin/bash
# usage: tools/ab_gather.sh <workload> [steps] -- per-stage times of the gather kernels, one box
wl=${1:-c2}; steps=${2:-10}
for cfg in "PB_GATHER=staged" "PB_GATHER=rounds PB_ROUNDS_GEOM=0" "PB_GATHER=rounds PB_ROUNDS_GEOM=1"; do
  echo "== $cfg"
  env $cfg python tools/bench_stages.py $wl $steps 2>&1 | grep -v amdgpu.ids
done
