#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r03
python tools/pmc_traffic.py c2 > gpurun_out/pmc_traffic.log 2>&1 || { tail -5 gpurun_out/pmc_traffic.log; exit 1; }
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_driverflags.json 2> gpurun_out/${tag}_bench_c2_driverflags.err || exit 1
{ for n in 2 4 8; do python tools/bench_wshard.py $n c2 3; done; python tools/bench_wshard.py 8 c2-1e6 3; python tools/bench_wshard.py 8 c2 2; python tools/bench_wshard.py 8 c2-1e6 2; } 2>&1 | grep shard > gpurun_out/${tag}_wshard.log
{ python tools/bench_rank_rccl.py 8 c2 3; python tools/bench_rank_rccl.py 4 c2 3; python tools/bench_rank_rccl.py 2 c2 3; } 2>&1 | grep "rank " > gpurun_out/${tag}_rank_rccl.log
python bench.py --workload c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_nocpu.json 2>/dev/null
python -c "
import json
for f in ('r03_bench_c2', 'r03_bench_c2_driverflags'):
    d = json.load(open('gpurun_out/%s.json' % f)); print(f, d['value'], d['ms_per_step'], d['roofline']['traffic'], d['config'].get('cold_value'))
d = json.load(open('gpurun_out/r03_bench_c5_nocpu.json')); print('c5', d['value'], d['config']['one_pass']['evals_per_s'])
"
cat gpurun_out/${tag}_wshard.log gpurun_out/${tag}_rank_rccl.log
