# usage: tools/sweep_split.sh <workload> -- staged-kernel tiling (PB_STAGE_S, PB_STAGE_SPLIT) at N=1 and at rank sizes
wl=${1:-c2}
for cfg in "2 1" "2 2" "1 1" "1 2"; do set -- $cfg; echo "N=1 S=$1 split=$2: $(PB_STAGE_S=$1 PB_STAGE_SPLIT=$2 python tools/bench_stages.py $wl 6)"; done
for n in 2 4 8; do
echo "N=$n default: $(python tools/bench_rank.py $n $wl | head -1)"
for cfg in "2 1" "2 2" "2 4" "2 8" "1 2" "1 4" "1 8"; do set -- $cfg; echo "N=$n S=$1 split=$2: $(PB_STAGE_S=$1 PB_STAGE_SPLIT=$2 python tools/bench_rank.py $n $wl | head -1)"; done
done
