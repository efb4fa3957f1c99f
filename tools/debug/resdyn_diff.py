import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from pyratbay_amd import engine as eng, synth
case = synth.lbl_case(3001, 8, 6000, wnosamp=24, nlor=16, ndop=8, extent=60.0, cutoff=3.0, niso=2, seed=23, resolution=50000.0)
g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], g['wnosamp'], 2)
ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], len(iso['isomass']), g['own'])
lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'], case['ethresh'], resolution=True, max_layers=8)
temp, dens, isoz = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
want = lbl.extinction(temp, dens, isoz).cpu().numpy()
of, _ = lbl.last_state(8, 1)
lbl.set_gather_mode('dynamic')
for rep in range(3):
    got = lbl.extinction(temp, dens, isoz).cpu().numpy()
    print('rep', rep, 'ofactor', of)
    for l in range(8):
        d = np.abs(got[l, 0] - want[l, 0]) / np.abs(want[l, 0]).max()
        bad = np.flatnonzero(d > 1e-12)
        print(' layer', l, 'bad', len(bad), (bad[:5], bad[-5:]) if len(bad) else '', 'max', d.max())
