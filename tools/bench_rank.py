"""Per-rank compute time of the layer-sharded mode without the collectives (one GPU):
usage: python tools/bench_rank.py <world> [workload]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pyratbay_amd import engine, synth
from pyratbay_amd.dist import LayerShardedTransit

world = int(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
w = bench.WORKLOADS[name]
case = synth.lbl_case(w['nwave'], w['nlayers'], w['nlines'], wnstep=w['wnstep'],
                      niso=w['niso'], seed=42)
res = []
for rank in (0, world - 1):
    sh = LayerShardedTransit(case, world, rank)
    sh.world = 1                       # no process group here: the exchange becomes a copy
    m = sh.model
    n = len(sh.layers)
    wcols = sh.gather.pad
    ec_cols = torch.rand((sh.nlayers, wcols), dtype=torch.float64, device='cuda') * 1e-9
    for _ in range(3):
        m.lbl.extinction(sh.temp, sh.dens, sh.isoz, add=True, out=sh.ec[:n])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t = np.zeros(2)
    for _ in range(10):
        ev[0].record()
        m.lbl.extinction(sh.temp, sh.dens, sh.isoz, add=True, out=sh.ec[:n])
        ev[1].record()
        engine.transit_spectrum(ec_cols, m.raypath, m.radius, m.rstar, 0, sh.nlayers, 10.0)
        ev[2].record()
        torch.cuda.synchronize()
        t += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])]
    t /= 10
    # wall clock of back-to-back steps (host enqueue included), and of the enqueue alone
    import time
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for _ in range(50):
        m.lbl.extinction(sh.temp, sh.dens, sh.isoz, add=True, out=sh.ec[:n])
        engine.transit_spectrum(ec_cols, m.raypath, m.radius, m.rstar, 0, sh.nlayers, 10.0)
    w1 = time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    res.append((rank, n, t[0], t[1], m.lbl.last_gather_kernel, (w2 - w0) / 50 * 1e3,
                (w1 - w0) / 50 * 1e3))
    del sh
for r in res:
    print(f'world {world} rank {r[0]}: {r[1]} layers  extinction {r[2]:.3f} ms  '
          f'depth+spectrum on W/{world} {r[3]:.3f} ms  [{r[4]}]  '
          f'wall {r[5]:.3f} ms/step (host enqueue {r[6]:.3f})')
