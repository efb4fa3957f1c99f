#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <python script + args...> -- rocprofv3 kernel stats, top kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python "$@" > $out/run.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:12]:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
