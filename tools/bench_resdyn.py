"""`resolution` mode: the direct gather (k_ext_linterp) against the per-layer dynamic grids
(gather mode 'dynamic').  Compares the extinction of both and times them.
usage: python tools/bench_resdyn.py [workload=c2-res] [steps=5] [nlines] [nwave] [nlayers]"""
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pyratbay_amd import engine

name = sys.argv[1] if len(sys.argv) > 1 else 'c2-res'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = dict(bench.WORKLOADS[name])
if len(sys.argv) > 3:
    w['nlines'] = int(sys.argv[3])
if len(sys.argv) > 4:
    w['nwave'] = int(sys.argv[4])
if len(sys.argv) > 5:
    w['nlayers'] = int(sys.argv[5])
case = bench.make_case(w)
model = engine.LBLSpectrum(case, rt_path=w.get('rt_path', 'transit'))
out = {}
for mode in ('auto', 'dynamic'):
    model.lbl.set_gather_mode(mode)
    t0 = time.perf_counter()
    model.extinction()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    for _ in range(2):
        model.extinction()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.extinction()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[mode] = model.ec.clone()
    print(f'{name} {mode:8s}: {model.lbl.last_gather_kernel}: first call {first*1e3:.1f} ms, '
          f'then {dt*1e3:.3f} ms per extinction', flush=True)
a, b = out['auto'], out['dynamic']
scale = a.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
rel = ((a - b).abs() / scale).amax().item()
nz = bool(((a == 0) == (b == 0)).all())
print(f'largest difference / row maximum: {rel:.3e}; identical zero pattern: {nz}')
model.lbl.set_gather_mode('dynamic')
model.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'{name} dynamic: {dt*1e3:.3f} ms per spectrum ({1/dt:.1f} spectra/s)')
# host time of one call (the launches of every run are issued by one thread)
torch.cuda.synchronize()
hs = []
for _ in range(steps):
    t0 = time.perf_counter()
    model.extinction()
    hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f'{name} dynamic: host time of the call {np.median(hs)*1e3:.3f} ms (then the stream drains)')
