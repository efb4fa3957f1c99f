#!/bin/bash
# usage: tools/pmc_fetch.sh <outdir> <workload> -- FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes only
out=$1; wl=${2:-c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python tools/bench_stages.py $wl 3 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; }
done
python tools/pmc_summary.py $out | grep -A8 "staged\|combine"
