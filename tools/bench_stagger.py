import os, sys, time
sys.path.insert(0, '.')
import torch, bench
from pyratbay_amd import engine
case = bench.make_case(bench.WORKLOADS['c2'])
pipe = engine.SpectrumPipeline(case, depth=2, rt_path='transit')
for _ in range(6):
    pipe.submit()
pipe.flush(); torch.cuda.synchronize()
def run(K, stagger_cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stagger_cycles:
        with torch.cuda.stream(pipe.streams[1]):
            torch.cuda._sleep(stagger_cycles)
    for _ in range(K):
        pipe.submit()
    pipe.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for K in (10, 20, 50):
    for cyc in (0, 500000, 1000000, 1500000):
        ts = [run(K, cyc) for _ in range(5)]
        print(f'K={K} stagger {cyc:8d} cycles: {min(ts):.3f} ms total, {min(ts)/K:.4f} ms/step')
    pipe.count = 0
