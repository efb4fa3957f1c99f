"""Per-stage device timing of the hot path (torch events on the current stream).
usage: python tools/bench_stages.py [workload] [steps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine, synth

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = bench.WORKLOADS[name]
case = bench.make_case(w)
nshard = int(sys.argv[3]) if len(sys.argv) > 3 else 1
from pyratbay_amd.dist import shard_bounds
b = shard_bounds(case['grid']['nwave'], nshard)
r = nshard // 2
model = engine.LBLSpectrum(case, rt_path=w.get('rt_path', 'transit'), wbegin=int(b[r]),
                           wcount=int(b[r + 1] - b[r]))
if os.environ.get('PB_BENCH_GATHER'):
    model.lbl.set_gather_mode(os.environ['PB_BENCH_GATHER'])
for _ in range(2):
    model.run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = [0.0, 0.0, 0.0]
for _ in range(steps):
    ev[0].record(); model.extinction()
    ev[1].record(); model.optical_depth()
    ev[2].record(); model.rt()
    ev[3].record()
    torch.cuda.synchronize()
    for i in range(3):
        tot[i] += ev[i].elapsed_time(ev[i + 1])
print(f'{name} shard 1/{nshard} [{model.lbl.last_gather_kernel}]: extinction {tot[0]/steps:.3f} ms  odepth {tot[1]/steps:.3f} ms  '
      f'spectrum {tot[2]/steps:.3f} ms  total {sum(tot)/steps:.3f} ms '
      f'({steps/sum(tot)*1e3:.1f} spectra/s)')

if os.environ.get('PB_GRAPH') == '1':
    replay = model.capture()
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps * 5):
        replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps * 5)
    t0 = time.perf_counter()
    for _ in range(steps * 5):
        model.run()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / (steps * 5)
    print(f'{name}: eager {de*1e3:.3f} ms/step, HIP graph replay {dt*1e3:.3f} ms/step')
