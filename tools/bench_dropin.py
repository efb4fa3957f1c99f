"""Level-1 drop-in (pyratbay.lib._extcoeff.extinction, one layer per call, HOST arrays in and
out): seconds per spectrum at C2, everything the call does on the host included (content hashes
of the cached arrays, the ext row over PCIe both ways).  usage: python tools/bench_dropin.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                           # noqa: E402


def main():
    from pyratbay_amd import engine
    from pyratbay_amd.lib import _extcoeff
    case = bench.make_case(bench.WORKLOADS['c2'])
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                                 g['wnosamp'], True)
    profile, psize, pindex = vt.flat(), np.array(vt.size), np.array(vt.index)
    vt.close()
    nl, nw = atm['nlayers'], g['nwave']
    ec = np.zeros((nl, 1, nw))

    def spectrum():
        for k in range(nl):
            _extcoeff.extinction(ec[k], profile, psize, pindex, vg['lorentz'], vg['doppler'],
                                 g['wn'], g['own'], g['divisors'], atm['dens'][k],
                                 atm['mol_radius'], atm['mol_mass'], iso['isoimol'],
                                 iso['isomass'], iso['isoratio'], iso['isoz'][:, k].copy(),
                                 iso['isoiext'], ln['lwn'], ln['elow'], ln['gf'], ln['lid'],
                                 vg['cutoff'], case['ethresh'], float(atm['temp'][k]), 0, 1, 0)

    t0 = time.perf_counter()
    spectrum()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    spectrum()
    again = time.perf_counter() - t0
    print(f'C2 through the per-layer drop-in: first pass {first:.2f} s (uploads the table and the '
          f'line list), steady state {again:.3f} s per 80-layer extinction = '
          f'{again / nl * 1e3:.1f} ms per call')


if __name__ == '__main__':
    main()
