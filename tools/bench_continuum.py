"""Device time and streamed bytes of the fused continuum pass at the C2 shape.
usage: python tools/bench_continuum.py [nlayers] [nwave]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyratbay_amd import continuum as ct, engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 80
W = int(sys.argv[2]) if len(sys.argv) > 2 else 100001
engine.require_gpu()
rng = np.random.default_rng(1)
wn = np.linspace(4000.0, 9000.0, W)
pressure = np.logspace(-6, 2, L)
temp = np.linspace(1000.0, 1700.0, L)
ntot = pressure * ct.BAR / (ct.K * temp)
dens = {'H2': 0.85 * ntot, 'He': 0.149 * ntot, 'H': 1e-3 * ntot, 'e-': 1e-7 * ntot,
        'Na': 2e-6 * ntot, 'K': 1e-7 * ntot}
temps = np.array([60, 100, 150, 200, 250, 300, 350, 400, 500, 600, 700, 800, 900, 1000, 2000,
                  3000, 4000, 5000, 6000, 7000], float)
cia = []
for species in (['H2', 'H2'], ['H2', 'He']):
    tab = 10**rng.uniform(-8, -5, (len(temps), 64))
    cia.append(ct.Collision_Induced(table=(tab, species, temps, np.linspace(3000, 10000, 64)),
                                    wn=wn))
lec = ct.Lecavelier(pressure, wn=wn)
models = [ct.Kurucz(wn, s) for s in ('H', 'He', 'H2', 'e-')] + [lec] + cia + [ct.Hydrogen_Ion(wn)]
fused = ct.Continuum(wn, pressure, models)
alk = ct.Continuum(wn, pressure, [ct.SodiumVdW(pressure, wn=wn), ct.PotassiumVdW(pressure, wn=wn)])
ec = torch.zeros((L, W), dtype=torch.float64, device='cuda')
for name, c, nbytes in (('fused (5 rank-1 + 2 CIA + H-)', fused,
                         16.0 * L * W + 2 * 2 * 8.0 * L * W + (5 + 7) * 8.0 * W),
                        ('alkali Na + K', alk, 2 * 16.0 * L * W)):
    for _ in range(3):
        c.add(ec, temp, dens)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        c.add(ec, temp, dens)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'{name}: {ms:.3f} ms per call incl. host staging of the per-layer factors; '
          f'{nbytes / 1e6:.0f} MB streamed -> {nbytes / ms / 1e6:.0f} GB/s')
