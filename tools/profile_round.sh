#!/bin/bash
# usage: tools/profile_round.sh <tag> -- the round's evidence in one GPU call:
#   * the default bench JSON (un-profiled)                           -> gpurun_out/<tag>_bench_c2.json
#   * the driver's flags (--steps 20 --warmup 5)                     -> gpurun_out/<tag>_bench_c2_driverflags.json
#   * rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_kernel_stats.csv
#   * bench JSONs of the other workloads (c2-res, c3, c4, c5)        -> gpurun_out/<tag>_bench_*.json
#   * rocprofv3 kernel stats of c2-res and c5                        -> gpurun_out/<tag>_kernel_stats_{c2res,c5}.csv
#   * per-rank times of the wavenumber decomposition                 -> gpurun_out/<tag>_wshard.log
#   * the same with the collectives through a one-rank RCCL group    -> gpurun_out/<tag>_rank_rccl.log
#   * retrieval batch: one pass against two passes                   -> gpurun_out/<tag>_table_transit_ab.log
#   * PMC traffic of the dominant kernel keyed by the library hash   -> gpurun_out/pmc_traffic.json
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_driverflags.json 2> gpurun_out/${tag}_bench_c2_driverflags.err || exit 1
echo "bench c2 done"
rm -rf gpurun_out/prof_bench gpurun_out/prof_c2res gpurun_out/prof_c5
# one spectrum at a time (PB_STREAMS=1): the kernel durations then agree with roofline.kernel_ms,
# which bench.py measures in its un-pipelined pass
PB_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --no-cpu-baseline --sustain-seconds 0 > gpurun_out/prof_bench.log 2>&1 || exit 1
cp $(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
PB_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2res -- python bench.py --workload c2-res --steps 10 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/prof_c2res.log 2>&1 || exit 1
cp $(find gpurun_out/prof_c2res -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats_c2res.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python bench.py --workload c5 --steps 20 --no-cpu-baseline > gpurun_out/prof_c5.log 2>&1 || exit 1
cp $(find gpurun_out/prof_c5 -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats_c5.csv
rm -rf gpurun_out/prof_marker
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d gpurun_out/prof_marker -- python tools/show_timestamps.py c2 > gpurun_out/${tag}_timestamps.log 2>&1 || exit 1
cp $(find gpurun_out/prof_marker -name '*marker_api_stats.csv' | head -1) gpurun_out/${tag}_marker_api_stats.csv
echo "kernel stats done"
python bench.py --workload c5 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || exit 1
PB_TRANSIT_MFMA=0 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/${tag}_bench_c5_vector.json 2> gpurun_out/${tag}_bench_c5_vector.err || exit 1
python bench.py --workload c2-res > gpurun_out/${tag}_bench_c2res.json 2> gpurun_out/${tag}_bench_c2res.err || exit 1
python bench.py --workload c3 --steps 5 --warmup 2 --cpu-layers 4 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err || exit 1
python bench.py --workload c4 --steps 3 --warmup 1 --cpu-layers 2 > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err || exit 1
echo "other workloads done"
{ for n in 2 4 8; do python tools/bench_wshard.py $n c2 3; done; python tools/bench_wshard.py 8 c2-1e6 3; python tools/bench_wshard.py 8 c2 2; python tools/bench_wshard.py 8 c2-1e6 2; } 2>&1 | grep shard > gpurun_out/${tag}_wshard.log
python tools/bench_dropin.py 2>&1 | grep drop-in > gpurun_out/${tag}_dropin.log
python tools/bench_outofcore.py --lines 1e8 --budget-gib 16 --check --out gpurun_out/${tag}_outofcore_1e8.json > gpurun_out/${tag}_outofcore.log 2>&1 || { tail -3 gpurun_out/${tag}_outofcore.log; exit 1; }
python tools/bench_tt.py > gpurun_out/${tag}_table_transit_ab.log 2>&1 || { tail -3 gpurun_out/${tag}_table_transit_ab.log; exit 1; }
{ python tools/bench_rank_rccl.py 8 c2 3; python tools/bench_rank_rccl.py 4 c2 3; python tools/bench_rank_rccl.py 2 c2 3; } 2>&1 | grep "rank " > gpurun_out/${tag}_rank_rccl.log
echo "rank shards done"
python tools/pmc_traffic.py c2 > gpurun_out/pmc_traffic.log 2>&1 || { tail -5 gpurun_out/pmc_traffic.log; exit 1; }
tail -2 gpurun_out/pmc_traffic.log
