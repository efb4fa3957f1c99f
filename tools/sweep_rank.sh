# per-rank compute of the layer-sharded mode (one GPU), both line-list sizes
for wl in c2 c2-1e6; do
  python tools/bench_stages.py $wl 6
  for n in 2 4 8; do python tools/bench_rank.py $n $wl | head -1; done
done
python tools/bench_stages.py c3 3
