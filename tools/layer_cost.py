"""Where the extinction time goes: per block of layers, gather time in both kernels.
usage: python tools/layer_cost.py [workload] [layers per block]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pyratbay_amd import engine, synth

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = bench.WORKLOADS[name]
case = synth.lbl_case(w['nwave'], w['nlayers'], w['nlines'], wnstep=w['wnstep'],
                      niso=w['niso'], seed=42)
model = engine.LBLSpectrum(case, rt_path='transit')
model.run()
L = w['nlayers']
ofac, _ = model.lbl.last_state(L, 1)
resident, block = model.lbl.last_layer_kinds(L)
print('resident layers:', resident.tolist())
print('block sizes (doubles):', block.tolist())
temp, dens, isoz = model.temp, model.dens, model.isoz
vt = model.voigt
print('voigt sizes (idop 0, every 10th ilor):', np.asarray(vt.size)[0, ::10] if hasattr(vt, 'size') else '')
for l0 in range(0, L, step):
    l1 = min(L, l0 + step)
    res = []
    for mode in ('staged', 'auto'):
        model.lbl.set_gather_mode(mode)
        rep = L // (l1 - l0)       # a full-size launch made of copies of this block
        t, d, z = (temp[l0:l1].repeat(rep).contiguous(), dens[l0:l1].repeat(rep, 1).contiguous(),
                   isoz[:, l0:l1].repeat(1, rep).contiguous())
        for _ in range(2):
            model.lbl.extinction(t, d, z)
        torch.cuda.synchronize()
        model.lbl.timing_begin(16)
        for _ in range(5):
            model.lbl.extinction(t, d, z)
        torch.cuda.synchronize()
        ms, n = model.lbl.timing_end()
        res.append((ms / n, model.lbl.last_gather_kernel))
    print(f'layers {l0:2d}-{l1-1:2d} ofactor {ofac[l0]:3d}..{ofac[l1-1]:3d}  '
          f'staged {res[0][0]:.3f} ms  auto {res[1][0]:.3f} ms ({res[1][1]})')
