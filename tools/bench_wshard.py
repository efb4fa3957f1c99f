"""Per-rank compute of the WAVENUMBER decomposition without the collectives (one GPU): a middle
shard of `world`, one-call form (records + maxima of every group on every rank) against the
two-phase form (records of the shard's groups only; the all-reduce of the maxima is left out).
usage: python tools/bench_wshard.py <world> [workload]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine
from pyratbay_amd.dist import shard_bounds

world = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
case = bench.make_case(bench.WORKLOADS[name])
b = shard_bounds(case['grid']['nwave'], world)
r = world // 2
m = engine.LBLSpectrum(case, rt_path='transit', wbegin=int(b[r]), wcount=int(b[r + 1] - b[r]))
for label, exch in (('one call', None), ('two-phase', lambda t: None)):
    m.kmax_exchange = exch
    for _ in range(3):
        m.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        m.run()
    torch.cuda.synchronize()
    print(f'{name} shard {r}/{world} ({label}): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step '
          f'[{m.lbl.last_gather_kernel}]', flush=True)
