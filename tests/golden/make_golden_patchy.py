#!/usr/bin/env python3
"""Generate tests/golden/g11_patchy.npz: the reference's patchy-cloud cases.

Build container only (needs /root/reference; imports the package the way
make_golden_e2e.py does):

    python tests/golden/make_golden_patchy.py

The reference's tests/test_{transmission,emission,eclipse}.py::test_*_patchy run the full
test configuration with `fpatchy = 0.5` and `clouds = deck -3.0 / lecavelier 10.0 -15.0`.
Their expected files include a sampled cross-section table that needs a HITRAN download, so
the cases are run here WITHOUT `sampled_cross_sec` (CIA, alkali, Rayleigh and the clouds stay)
and the fixture is what the reference computes for that: the two extinction arrays the
package hands to its optical-depth step (`opacity.ec`, `opacity.ec_cloud`), every 8th
column, and the spectra it returns (`spec.spectrum`, `spec.clear`, `spec.cloudy`) for
fpatchy = 0.5, plus the run with fpatchy = 0 and 1 the reference's tests also make.
Only data is written.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e                      # noqa: E402
import make_golden_reference_cases as rc           # noqa: E402

REF = e2e.REF
STEP = 8


def main():
    work = tempfile.mkdtemp(prefix='pb_patchy_')
    store = {}
    try:
        pb = e2e.reference_package(work)
        os.chdir(work)
        reset = {'fpatchy': '0.5', 'clouds': 'deck -3.0\nlecavelier 10.0 -15.0'}
        for rt in ('transit', 'emission', 'eclipse'):
            cfg = rc.make_config(work, f'{REF}/tests/configs/{rc.BASE[rt]}', reset,
                                 ['sampled_cross_sec'], f'patchy_{rt}')
            pyrat = pb.run(cfg)
            spec, atm, od, opa = pyrat.spec, pyrat.atm, pyrat.od, pyrat.opacity
            assert opa.is_patchy
            cols = np.arange(0, spec.nwave, STEP)
            deck = [m for m in opa.models if getattr(m, 'name', '') == 'deck'][0]
            store[f'{rt}_wn'] = spec.wn[cols]
            store[f'{rt}_ec'] = opa.ec[:, cols]
            store[f'{rt}_ec_cloud'] = opa.ec_cloud[:, cols]
            store[f'{rt}_spectrum'] = spec.spectrum[cols]
            store[f'{rt}_clear'] = spec.clear[cols]
            store[f'{rt}_cloudy'] = spec.cloudy[cols]
            store[f'{rt}_ideep'] = np.asarray(od.ideep)[cols]
            store[f'{rt}_ideep_clear'] = np.asarray(od.ideep_clear)[cols]
            store[f'{rt}_deck'] = np.array([deck.pars[0], deck.itop, deck.rsurf, deck.tsurf])
            store[f'{rt}_scalars'] = np.array(
                [atm.rtop, np.nan if atm.rstar is None else atm.rstar, od.maxdepth,
                 atm.rplanet, opa.fpatchy], float)
            for key, val in (('press', atm.press), ('temp', atm.temp), ('radius', atm.radius)):
                store[f'{rt}_{key}'] = val
            if rt != 'transit':
                store[f'{rt}_mu'] = spec.quadrature_mu
                store[f'{rt}_weights'] = np.ravel(spec.quadrature_weights)
            if rt == 'eclipse':
                store[f'{rt}_starflux'] = spec.starflux[cols]
            # the reference's tests then set fpatchy = 0 and 1 and run again
            for f in (0.0, 1.0):
                pyrat.opacity.fpatchy = f
                pyrat.run()
                store[f'{rt}_spectrum_f{int(f)}'] = pyrat.spec.spectrum[cols].copy()
            np.testing.assert_allclose(store[f'{rt}_spectrum_f0'], store[f'{rt}_clear'])
            np.testing.assert_allclose(store[f'{rt}_spectrum_f1'], store[f'{rt}_cloudy'])
            print(rt, 'W', spec.nwave, '->', len(cols), 'columns; L', atm.nlayers,
                  'deck itop', deck.itop)
        np.savez_compressed(os.path.join(HERE, 'g11_patchy.npz'), **store)
    finally:
        os.chdir(HERE)
        shutil.rmtree(work, ignore_errors=True)
    print('g11_patchy.npz', os.path.getsize(os.path.join(HERE, 'g11_patchy.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
