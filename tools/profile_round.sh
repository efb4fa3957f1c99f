#!/bin/bash
# usage: tools/profile_round.sh <tag> <part> -- the round's evidence, to be run after the last kernel
# change (everything is keyed to the library that is in the tree: its hash goes into
# <tag>_library.txt and pmc_traffic.json).  Three GPU calls (a gpurun call is limited to 20 minutes):
#   part a: the bench lines and kernel stats (~8 min)   part b: counters and traffic (~9 min)
#   part fuzz: the randomised parity campaign (~8 min)
# Writes gpurun_out/<tag>_*; tools/keep_evidence.sh <tag> copies what is kept into profiles/.
#
#   <tag>_library_<part>.txt          sha256 of pyratbay_amd/libpbhip.so as that part ran it
#   <tag>_bench_c2.json               default bench (the driver's command), un-profiled
#   <tag>_bench_c2_driverflags.json   --steps 20 --warmup 5
#   <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of the default bench, one at a time
#   <tag>_bench_{c2res,c3,c4,c5,c5em,c2bands,c3bands}.json   the other workloads
#   <tag>_kernel_stats_{c5,c5em}.csv  rocprofv3 kernel stats of the retrieval batches
#   <tag>_wshard.log, <tag>_rank_rccl.log   per-rank times of the wavenumber decomposition
#   <tag>_ordered.log                 retrieval batch: transit kernels in grid / depth order, c5 and c5-emission with the ordering off and on
#   <tag>_tile_times_c2bands.log      per-tile time spread of a band-structured list (tools/tile_times.py)
#   <tag>_pmc_c2.json                 SQ / cache counters of the C2 kernels (tools/pmc.sh)
#   <tag>_write_size_probe.csv        WRITE_SIZE of 8- and 16-byte-per-lane stores (tools/write_size_probe.hip)
#   pmc_traffic.json                  FETCH_SIZE / WRITE_SIZE of the dominant kernels, every workload
#   <tag>_fuzz_all.log                (with `fuzz`) the randomised parity campaign
tag=${1:-r05}
part=${2:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
sha256sum pyratbay_amd/libpbhip.so > gpurun_out/${tag}_library_${part}.txt
if [ "$part" = "a" ]; then
python bench.py > gpurun_out/${tag}_bench_c2.json 2> gpurun_out/${tag}_bench_c2.err || exit 1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_c2_driverflags.json 2> gpurun_out/${tag}_bench_c2_driverflags.err || exit 1
echo "bench c2 done"
rm -rf gpurun_out/prof_bench gpurun_out/prof_c5 gpurun_out/prof_c5em
# one spectrum at a time (PB_STREAMS=1): the kernel durations then agree with roofline.kernel_ms,
# which bench.py measures in its un-pipelined pass
PB_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --no-cpu-baseline --no-rank-projection --no-legs --sustain-seconds 0 > gpurun_out/prof_bench.log 2>&1 || exit 1
cp $(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python bench.py --workload c5 --steps 20 --no-cpu-baseline > gpurun_out/prof_c5.log 2>&1 || exit 1
cp $(find gpurun_out/prof_c5 -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats_c5.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5em -- python bench.py --workload c5-emission --steps 20 --no-cpu-baseline > gpurun_out/prof_c5em.log 2>&1 || exit 1
cp $(find gpurun_out/prof_c5em -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats_c5em.csv
echo "kernel stats done"
python bench.py --workload c5 > gpurun_out/${tag}_bench_c5.json 2> gpurun_out/${tag}_bench_c5.err || exit 1
python bench.py --workload c5-emission --steps 40 > gpurun_out/${tag}_bench_c5em.json 2> gpurun_out/${tag}_bench_c5em.err || exit 1
python bench.py --workload c2-res > gpurun_out/${tag}_bench_c2res.json 2> gpurun_out/${tag}_bench_c2res.err || exit 1
python bench.py --workload c2-bands --no-north-star > gpurun_out/${tag}_bench_c2bands.json 2> gpurun_out/${tag}_bench_c2bands.err || exit 1
python bench.py --workload c3 --steps 5 --warmup 2 --cpu-layers 4 > gpurun_out/${tag}_bench_c3.json 2> gpurun_out/${tag}_bench_c3.err || exit 1
python bench.py --workload c3-bands --steps 5 --warmup 2 --cpu-layers 4 > gpurun_out/${tag}_bench_c3bands.json 2> gpurun_out/${tag}_bench_c3bands.err || exit 1
python bench.py --workload c4 --steps 3 --warmup 1 --cpu-layers 2 > gpurun_out/${tag}_bench_c4.json 2> gpurun_out/${tag}_bench_c4.err || exit 1
echo "other workloads done"
{ for n in 2 4 8; do python tools/bench_wshard.py $n c2 3; done; python tools/bench_wshard.py 8 c2-1e6 3; } 2>&1 | grep shard > gpurun_out/${tag}_wshard.log
{ python tools/bench_rank_rccl.py 8 c2 3 3; python tools/bench_rank_rccl.py 8 c2 3; python tools/bench_rank_rccl.py 4 c2 3 3; python tools/bench_rank_rccl.py 2 c2 3 3; python tools/bench_rank_rccl.py 8 c2-1e6 3 3; } 2>&1 | grep "rank " > gpurun_out/${tag}_rank_rccl.log
python tools/bench_dropin.py 2>&1 | grep drop-in > gpurun_out/${tag}_dropin.log
echo "rank shards done"
{ python tools/bench_ordered.py; for o in 0 1 0 1; do echo "PB_COLUMN_ORDER=$o:"; PB_COLUMN_ORDER=$o python bench.py --workload c5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' c5', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms per 64 walkers')"; done; for o in 0 1; do echo "PB_COLUMN_ORDER=$o:"; PB_COLUMN_ORDER=$o python bench.py --workload c5-emission --steps 40 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' c5-emission', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms per 64 walkers')"; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ordered.log
python tools/tile_times.py c2-bands 5 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_tile_times_c2bands.log
echo "ordered batches and tile times done"
fi
if [ "$part" = "b" ]; then
bash tools/pmc.sh gpurun_out/pmc_${tag} c2 > gpurun_out/pmc_${tag}.log 2>&1
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open(f'gpurun_out/pmc_{tag}/pmc_summary.json'))
lib = open(f'gpurun_out/{tag}_library_b.txt').read().split()[0]
keep = {k.replace('void ', ''): v for k, v in d.items()
        if any(s in k for s in ('k_ext_', 'k_records', 'k_transit', 'k_layer_state', 'k_path_blocks'))}
json.dump({'library_sha256': lib, 'command': 'tools/pmc.sh (rocprofv3 --pmc, one counter set per pass) -- '
           'python tools/bench_stages.py c2 3', 'note': 'SQ_* wave counters are quad-cycles',
           'kernels': keep}, open(f'gpurun_out/{tag}_pmc_c2.json', 'w'), indent=1, sort_keys=True)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/write_size_probe.hip -o gpurun_out/write_size_probe > gpurun_out/write_size_probe_build.log 2>&1 && {
  rm -rf gpurun_out/prof_wsp
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_wsp -- ./gpurun_out/write_size_probe > gpurun_out/prof_wsp.log 2>&1
  cp $(find gpurun_out/prof_wsp -name '*counter_collection.csv' | head -1) gpurun_out/${tag}_write_size_probe.csv
  rm -f gpurun_out/write_size_probe
}
python tools/pmc_traffic.py c2 c2-1e6 c2-bands c3 c4 c5 c5-emission > gpurun_out/pmc_traffic.log 2>&1 || { tail -5 gpurun_out/pmc_traffic.log; exit 1; }
tail -2 gpurun_out/pmc_traffic.log
fi
if [ "$part" = "fuzz" ]; then
  bash tools/fuzz_all.sh 1 > gpurun_out/${tag}_fuzz_all.log 2>&1
  grep -c "exit 0" gpurun_out/${tag}_fuzz_all.log
fi
echo "evidence done"
