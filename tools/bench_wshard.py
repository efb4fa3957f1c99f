"""Per-rank compute of the WAVENUMBER decomposition without the collectives (one GPU): a middle
shard of `world`, one-call form (records + maxima of every group on every rank) against the
two-phase form (records of the shard's groups only; the all-reduce of the maxima is left out),
one spectrum at a time and with `streams` independent spectra in flight per rank
(engine.SpectrumPipeline on the shard).
usage: python tools/bench_wshard.py <world> [workload] [streams]"""
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
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(os.environ.get('PB_WSHARD_STEPS', '50'))
case = bench.make_case(bench.WORKLOADS[name])
b = shard_bounds(case['grid']['nwave'], world)
r = world // 2
kw = dict(rt_path='transit', wbegin=int(b[r]), wcount=int(b[r + 1] - b[r]),
          materialize_depth=os.environ.get('PB_NO_DEPTH') != '1')
m = engine.LBLSpectrum(case, **kw)
for label, exch in (('one call', None), ('two-phase', lambda t: None)):
    m.kmax_exchange = exch
    for _ in range(3):
        m.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.run()
    torch.cuda.synchronize()
    print(f'{name} shard {r}/{world} ({label}): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step '
          f'[{m.lbl.last_gather_kernel}]', flush=True)
if streams > 1:
    pipe = engine.SpectrumPipeline(case, depth=streams, voigt=m.voigt, lines=m.lines, **kw)
    for mm in pipe.models:
        mm.kmax_exchange = lambda t: None
    for _ in range(2 * streams):
        pipe.submit()
    pipe.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit()
    pipe.flush()
    torch.cuda.synchronize()
    print(f'{name} shard {r}/{world} (two-phase, {streams} in flight): '
          f'{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/spectrum '
          f'[{pipe.models[0].lbl.last_gather_kernel}]', flush=True)
