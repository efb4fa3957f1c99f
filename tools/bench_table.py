"""Table-based eval (BASELINE.json configs[4] shape: W=1e5, L=80, S=4 species, ntemp=10):
interp_ec -> transit optical depth -> transmission per eval, table resident in HBM.
usage: python tools/bench_table.py [steps]"""
import os
import sys
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyratbay_amd import engine, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S, ntemp, L, W = 4, 10, 80, 100001
atm = synth.synthetic_atmosphere(L, ('H2', 'He', 'H2O', 'CO', 'CO2', 'CH4'),
                                 (0.85, 0.149, 4e-4, 5e-4, 1e-7, 1e-4))
wn = 4000.0 + 0.05 * np.arange(W)
gen = torch.Generator(device='cuda').manual_seed(7)
ttable = np.linspace(300.0, 3000.0, ntemp)
# values 10**U(-30,-20), smooth in T
base = torch.rand((S, 1, L, W), generator=gen, device='cuda', dtype=torch.float64) * 10 - 30
slope = torch.rand((S, 1, L, W), generator=gen, device='cuda', dtype=torch.float64)
tt = torch.linspace(0, 1, ntemp, device='cuda', dtype=torch.float64).view(1, ntemp, 1, 1)
etable = torch.pow(10.0, base + slope * tt)
del base, slope
model = engine.TableSpectrum(etable, ttable, wn, atm['radius'], atm['rstar'])
rng = np.random.default_rng(7)
nwalk = 64
temps = engine.dev(atm['temp'][None, :] * rng.uniform(0.9, 1.1, (nwalk, 1)))
dens = engine.dev(atm['dens'][None, :, 2:6] * rng.uniform(0.9, 1.1, (nwalk, 1, S)))
for i in range(3):
    model.eval(temps[i], dens[i])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = np.zeros(3)
for i in range(steps):
    w = i % nwalk
    ev[0].record()
    engine.interp_ec(model.ec, model.etable, model.ttable, temps[w], dens[w], 0, L,
                     assign=True)
    ev[1].record()
    spec, depth, ideep = engine.transit_spectrum(model.ec, model.raypath, model.radius,
                                                 model.rstar, 0, L, 10.0)
    ev[2].record()
    ev[3].record()
    torch.cuda.synchronize()
    tot += [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
tot /= steps
bytes_interp = (16.0 * S + 8.0) * L * W
print(json.dumps({'workload': f'table eval: W={W} L={L} S={S} ntemp={ntemp}',
                  'interp_ec_ms': tot[0], 'odepth_ms': tot[1], 'spectrum_ms': tot[2],
                  'evals_per_s': 1e3 / tot.sum(),
                  'interp_ec_GBps': bytes_interp / tot[0] / 1e6,
                  'table_bytes': etable.numel() * 8}))
