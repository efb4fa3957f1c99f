"""compute_opacity (runmode = opacity): every (T, p) cell of a cross-section grid as one
batched extinction job.  usage: python tools/bench_opacity.py [ntemp] [workload] [gather mode]
(workload c2-res: a constant-resolving-power grid, gather mode 'dynamic' or 'auto' = direct)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pyratbay_amd import engine, synth, opacity_table as ot

ntemp = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
w = bench.WORKLOADS[name]
case = bench.make_case(w)
res = case['grid'].get('resolution') is not None
g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
nl = atm['nlayers']
vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                             g['wnosamp'], 2 if res else 0)
ll = engine.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], len(iso['isomass']), g['own'])
lbl = engine.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                 iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                 vg['cutoff'], case['ethresh'], resolution=res, max_layers=ntemp * nl)
if len(sys.argv) > 3:
    lbl.set_gather_mode(sys.argv[3])
tgrid = np.linspace(500.0, 2500.0, ntemp)
pf = np.stack([synth.partition_function(tgrid)] * len(iso['isomass']))
out = None
for _ in range(2):
    out = ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
cells = ntemp * nl
print(f'{name}: {ntemp} temperatures x {nl} layers = {cells} cells x {g["nwave"]} samples, '
      f'{len(ln["lwn"])} lines: {dt*1e3:.1f} ms per table ({dt/cells*1e6:.1f} us per cell, '
      f'{out.numel()*8/1e9:.2f} GB table; kernel {lbl.last_gather_kernel})')
