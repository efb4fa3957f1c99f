#!/bin/bash
# usage: tools/sweep_wshard.sh <world> <workload> -- tiling sweep of a middle wavenumber shard:
# sub-tiles per workgroup (PB_STAGE_S) x workgroups per tile (PB_STAGE_SPLIT), one spectrum at a
# time and two in flight (tools/bench_wshard.py)
world=${1:-8}; wl=${2:-c2}
for S in 1 2; do
  for split in 2 3 4 5 6 7 8; do
    echo "== S=$S split=$split"
    PB_STAGE_S=$S PB_STAGE_SPLIT=$split python tools/bench_wshard.py $world $wl 2 2>&1 | grep -E "two-phase"
  done
done
