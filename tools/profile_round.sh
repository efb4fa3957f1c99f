#!/bin/bash
# usage: tools/profile_round.sh <tag> -- the round's evidence in one GPU call:
#   * rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_kernel_stats.csv
#   * the default bench JSON (un-profiled)                           -> gpurun_out/<tag>_bench_c2.json
#   * bench JSONs of the other workloads                             -> gpurun_out/<tag>_bench_*.json
#   * PMC traffic of the dominant kernel keyed by the library hash   -> gpurun_out/pmc_traffic.json
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || exit 1
echo "bench c2 done"
rm -rf gpurun_out/prof_bench
# one spectrum at a time (PB_STREAMS=1): the kernel durations then agree with roofline.kernel_ms,
# which bench.py measures in its un-pipelined pass
PB_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 || exit 1
cp $(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
echo "kernel stats done"
python bench.py --workload c5 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || exit 1
python bench.py --workload c3 --steps 5 --warmup 2 --cpu-layers 4 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err || exit 1
python bench.py --workload c4 --steps 3 --warmup 1 --cpu-layers 2 > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err || exit 1
echo "other workloads done"
python tools/pmc_traffic.py c2 > gpurun_out/pmc_traffic.log 2>&1 || { tail -5 gpurun_out/pmc_traffic.log; exit 1; }
tail -2 gpurun_out/pmc_traffic.log
