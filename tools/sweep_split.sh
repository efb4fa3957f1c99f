for cfg in "2 1" "1 1" "2 2" "1 2" "2 4"; do set -- $cfg; echo "N=1 S=$1 split=$2: $(PB_STAGE_S=$1 PB_STAGE_SPLIT=$2 python tools/bench_stages.py c2-1e6 4)"; done
for n in 8 4; do
echo "N=$n default: $(python tools/bench_rank.py $n c2-1e6 | head -1)"
for cfg in "2 4" "2 8" "1 2" "1 4" "1 8"; do set -- $cfg; echo "N=$n S=$1 split=$2: $(PB_STAGE_S=$1 PB_STAGE_SPLIT=$2 python tools/bench_rank.py $n c2-1e6 | head -1)"; done
done
