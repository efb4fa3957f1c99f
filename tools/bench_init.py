"""Initialisation latency of the line-list side: co-add grouping + sorts + uploads
(pb_lines_create) and the plan (pb_lbl_create) at 1e6 and 1e7 lines on the C2 grid.
usage: python tools/bench_init.py [nlines ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyratbay_amd import engine, synth

for n in [int(float(x)) for x in (sys.argv[1:] or ['1e6', '1e7'])]:
    t0 = time.perf_counter()
    case = synth.lbl_case(100001, 80, n, wnstep=0.05, niso=1, seed=42)
    t_gen = time.perf_counter() - t0
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                                 g['wnosamp'])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ll = engine.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 1, g['own'])
    torch.cuda.synchronize()
    t_lines = time.perf_counter() - t0
    t0 = time.perf_counter()
    lbl = engine.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                     vg['cutoff'], 1e-30, max_layers=80)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    print(f'{n:>9d} lines: generate {t_gen:.2f} s | pb_lines_create {t_lines:.3f} s '
          f'({ll.ngroups} groups, {ll.nadd} co-added) | pb_lbl_create {t_plan:.3f} s',
          flush=True)
    del lbl, ll
