"""Throughput of consecutive spectra kept in flight on several HIP streams
(engine.SpectrumPipeline) against the one-stream loop, with the host's enqueue time per spectrum
(if the host cannot enqueue a spectrum faster than the device finishes one, more streams gain
nothing).  usage: python tools/bench_overlap.py [workload] [steps] [depths...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
depths = [int(x) for x in sys.argv[3:]] or [2, 3]
w = bench.WORKLOADS[name]
case = bench.make_case(w)
rt = w.get('rt_path', 'transit')
serial = engine.LBLSpectrum(case, rt_path=rt)
for _ in range(3):
    serial.run()
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(steps):
        serial.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: one stream   {1e3 * (t2 - t0) / steps:.3f} ms/spectrum  '
          f'(host enqueue {1e3 * (t1 - t0) / steps:.3f} ms)')
for depth in depths:
    pipe = engine.SpectrumPipeline(case, depth=depth, rt_path=rt, voigt=serial.voigt,
                                   lines=serial.lines)
    for _ in range(2 * depth):
        pipe.submit()
    pipe.flush()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit()
        t1 = time.perf_counter()
        pipe.flush()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{name}: {depth} streams    {1e3 * (t2 - t0) / steps:.3f} ms/spectrum  '
              f'(host enqueue {1e3 * (t1 - t0) / steps:.3f} ms)')
    del pipe
