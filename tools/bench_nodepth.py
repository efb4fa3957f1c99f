"""C2 with and without the optical-depth array materialised (LBLSpectrum(materialize_depth=...)):
one spectrum at a time and two in flight.  usage: python tools/bench_nodepth.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
case = bench.make_case(bench.WORKLOADS[name])
ref = None
for md in (True, False):
    m = engine.LBLSpectrum(case, rt_path='transit', materialize_depth=md)
    for _ in range(5):
        s = m.run()
    torch.cuda.synchronize()
    if ref is None:
        ref = s.clone()
    else:
        print('max rel diff of the spectrum without depth:', float((s / ref - 1).abs().max()))
    t0 = time.perf_counter()
    for _ in range(100):
        m.run()
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 100
    pipe = engine.SpectrumPipeline(case, depth=2, rt_path='transit', voigt=m.voigt, lines=m.lines,
                                   materialize_depth=md)
    for _ in range(32):
        pipe.submit()
    pipe.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        pipe.submit()
    pipe.flush(); torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / 200
    print(f'{name} materialize_depth={md}: one at a time {one * 1e3:.3f} ms, two in flight {two * 1e3:.3f} ms '
          f'= {1 / two:.0f} spectra/s; stages {m.timestamps}', flush=True)
