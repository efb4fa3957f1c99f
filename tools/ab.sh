#!/bin/bash
# usage: tools/ab.sh [workloads...] -- A/B of build_ab/libpbhip_{base,new}.so on one box
# (each copied over pyratbay_amd/libpbhip.so in turn, twice, stage timings printed)
wls=${@:-c2}
for rep in 1 2; do
  for v in base new; do
    cp build_ab/libpbhip_$v.so pyratbay_amd/libpbhip.so || exit 1
    for wl in $wls; do
      steps=20; [ $wl != c2 ] && steps=4
      echo -n "$v: "; python tools/bench_stages.py $wl $steps || exit 1
    done
  done
done
cp build_ab/libpbhip_new.so pyratbay_amd/libpbhip.so
