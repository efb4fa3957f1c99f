#!/bin/bash
# usage: tools/ab_multi.sh "v1 v2 ..." [workloads...] -- A/B of build_ab/libpbhip_<v>.so on one
# box (each copied over pyratbay_amd/libpbhip.so in turn, two rounds, stage timings printed);
# the library in place when the script starts is put back at the end.
variants=$1; shift
wls=${@:-c2}
cp pyratbay_amd/libpbhip.so /tmp/libpbhip_keep.so
for rep in 1 2; do
  for v in $variants; do
    cp build_ab/libpbhip_$v.so pyratbay_amd/libpbhip.so || exit 1
    for wl in $wls; do
      steps=20; [ $wl != c2 ] && steps=5
      echo -n "$v: "; python tools/bench_stages.py $wl $steps || exit 1
    done
  done
done
cp /tmp/libpbhip_keep.so pyratbay_amd/libpbhip.so
