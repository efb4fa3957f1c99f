import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from pyratbay_amd import synth
from pyratbay_amd.dist import LayerShardedTransit
w = bench.WORKLOADS['c2']
case = synth.lbl_case(w['nwave'], w['nlayers'], w['nlines'], wnstep=w['wnstep'], niso=w['niso'], seed=42)
sh = LayerShardedTransit(case, 1, 0)
for _ in range(5): sh.submit()
sh.flush(); torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): sh.submit()
t1 = time.perf_counter()
sh.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue per submit {1e3*(t1-t0)/n:.3f} ms; wall per spectrum {1e3*(t2-t0)/n:.3f} ms')
t0 = time.perf_counter()
for _ in range(n): sh.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue per step() {1e3*(t1-t0)/n:.3f} ms; wall per spectrum {1e3*(t2-t0)/n:.3f} ms')
