"""Retrieval batches (config 5) alternating on two HIP streams against the one-stream loop: the
interpolation of a batch is an HBM stream, its transit pass FP64 arithmetic -- do they overlap?
usage: python tools/bench_c5_overlap.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import bench_c5
from pyratbay_amd import engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
inp = bench_c5.inputs()
g, atm = inp['grid'], inp['atm']
models = [engine.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'], atm['rstar'])
          for _ in range(2)]
pb = engine.PassBands(g['wn'], inp['bands'])
batches = []
for b in range(4):
    temps, dens, radius = bench_c5.walkers(inp, bench_c5.BATCH, 700 + b)
    batches.append(tuple(engine.dev(x) for x in (temps, dens, radius)))
streams = [torch.cuda.Stream() for _ in range(2)]


def loop(nstream):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i in range(steps):
        j = i % nstream
        temps, dens, radius = batches[i % 4]
        with torch.cuda.stream(streams[j]):
            outs.append(models[j].eval_bands(temps, dens, pb, radius=radius, chunk=bench_c5.BATCH))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, outs


loop(1), loop(2)
for rep in range(2):
    one, o1 = loop(1)
    two, o2 = loop(2)
    same = all(torch.equal(a, b) for a, b in zip(o1, o2))
    print(f'c5: one stream {one:.3f} ms/batch ({bench_c5.BATCH / one * 1e3:.0f} evals/s), two streams '
          f'{two:.3f} ms/batch ({bench_c5.BATCH / two * 1e3:.0f} evals/s), {one / two:.3f}x, results equal: {same}')
