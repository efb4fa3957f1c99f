#!/bin/bash
# usage: tools/keep_evidence.sh <tag> -- copy the evidence of tools/profile_round.sh from gpurun_out/
# (scratch) into profiles/ (tracked): bench lines, kernel stats, counters, logs; pmc_traffic.json is
# the file bench.py reads (accepted only when its hash is the library's).
tag=${1:-r04}
cd "$(dirname "$0")/.."
for f in gpurun_out/${tag}_*.json gpurun_out/${tag}_*.csv gpurun_out/${tag}_*.log gpurun_out/${tag}_library_*.txt; do
  [ -f "$f" ] && cp "$f" profiles/
done
[ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
ls profiles | grep "^${tag}_" | wc -l
