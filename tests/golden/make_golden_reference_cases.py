#!/usr/bin/env python3
"""Generate tests/golden/g9_reference_cases.npz: the reference's OWN test cases that need no
downloaded line data -- tests/test_transmission.py, test_emission.py, test_eclipse.py
{clear, lecavelier, CIA, alkali, deck} -- run through the real package (imported as in
make_golden_e2e.py) from the reference's own config files with the same `remove` / `reset`
edits its tests apply, and checked HERE against the reference's expected_spectrum_*_test.npz
at the reference's tolerance (rtol 1e-4) before anything is stored:

    python tests/golden/make_golden_reference_cases.py

Stored per case: the reference's expected spectrum (its own golden vector), the spectrum of
this run, and the arrays the radiative-transfer stage needs (atmosphere, model parameters,
cloud-deck geometry, quadrature, stellar flux).  Shared: the grids and the two CIA tables
resampled to the model grid."""
import configparser
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e          # noqa: E402

REF = e2e.REF
BASE = dict(transit='spectrum_transmission_test.cfg', emission='spectrum_emission_test.cfg',
            eclipse='spectrum_eclipse_test.cfg')
EXPECT = dict(transit='transmission', emission='emission', eclipse='eclipse')
# (case, remove, reset) exactly as the reference's tests build them
CASES = {
    'transit': [
        ('clear', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali', 'clouds'], {}),
        ('lec', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'], {}),
        ('cia', ['sampled_cross_sec', 'alkali', 'clouds'], {}),
        ('alkali', ['sampled_cross_sec', 'continuum_cross_sec', 'clouds'],
         {'wl_low': '0.45 um', 'wl_high': '1.0 um'}),
        ('deck', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'], {'clouds': 'deck -3.0'}),
    ],
    'emission': [
        ('clear', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali', 'clouds'], {}),
        ('lec', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'],
         {'clouds': 'lecavelier 2.0 -4.0'}),
        ('cia', ['sampled_cross_sec', 'alkali', 'clouds'], {}),
        ('alkali', ['sampled_cross_sec', 'continuum_cross_sec', 'clouds'],
         {'wl_low': '0.45 um', 'wl_high': '1.0 um'}),
        ('deck', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'], {'clouds': 'deck -3.0'}),
    ],
    'eclipse': [
        ('lec', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'],
         {'clouds': 'lecavelier 2.0 -4.0'}),
        ('deck', ['sampled_cross_sec', 'continuum_cross_sec', 'alkali'], {'clouds': 'deck -1.0'}),
    ],
}


def make_config(work, cfile, reset, remove, tag):
    config = configparser.ConfigParser()
    config.optionxform = str
    with open(cfile) as f:
        text = f.read().replace('{ROOT}', REF + '/')
    config.read_string(text)
    config.set('pyrat', 'logfile', os.path.join(work, f'{tag}.log'))
    config.set('pyrat', 'ncpu', '1')
    config.set('pyrat', 'verb', '0')
    for var, val in reset.items():
        config.set('pyrat', var, val)
    for var in remove:
        config.remove_option('pyrat', var)
    path = os.path.join(work, f'{tag}.cfg')
    with open(path, 'w') as f:
        config.write(f)
    return path


def main():
    work = tempfile.mkdtemp(prefix='pb_refcases_')
    store = {}
    try:
        pb = e2e.reference_package(work)
        import pyratbay.constants as pc
        os.chdir(work)
        for rt, cases in CASES.items():
            for case, remove, reset in cases:
                tag = f'{rt}_{case}'
                cfg = make_config(work, f'{REF}/tests/configs/{BASE[rt]}', reset, remove, tag)
                pyrat = pb.run(cfg)
                spec, atm, od = pyrat.spec, pyrat.atm, pyrat.od
                if case == 'clear':
                    # the reference's analytic known-answer tests
                    if rt == 'transit':
                        expected = np.full(spec.nwave, (atm.radius[-1] / atm.rstar)**2)
                    else:
                        import pyratbay.spectrum as ps
                        expected = ps.bbflux(spec.wn, atm.temp[-1])
                else:
                    expected = np.load(f'{REF}/tests/expected/expected_spectrum_'
                                       f'{EXPECT[rt]}_{case}_test.npz')['arr_0']
                np.testing.assert_allclose(spec.spectrum, expected, rtol=1e-4)
                g = 'alk' if case == 'alkali' else 'std'
                if f'{g}_wn' not in store:
                    store[f'{g}_wn'] = spec.wn
                assert np.array_equal(store[f'{g}_wn'], spec.wn)
                for key, val in (('press', atm.press), ('temp', atm.temp), ('dens', atm.d),
                                 ('radius', atm.radius), ('species', np.array(atm.species))):
                    if key not in store:
                        store[key] = val
                    assert np.array_equal(store[key], val), (tag, key)
                store[f'{tag}_expected'] = expected
                store[f'{tag}_spectrum'] = spec.spectrum
                store[f'{tag}_ideep'] = np.asarray(od.ideep)
                store[f'{tag}_scalars'] = np.array(
                    [atm.rtop, np.nan if atm.rstar is None else atm.rstar, od.maxdepth,
                     atm.rplanet], float)
                if rt != 'transit':
                    store[f'{rt}_mu'] = spec.quadrature_mu
                    store[f'{rt}_weights'] = np.ravel(spec.quadrature_weights)
                if rt == 'eclipse':
                    store[f'{g}_starflux'] = spec.starflux
                    store[f'{tag}_fplanet'] = spec.fplanet
                for m, mtype in zip(pyrat.opacity.models, pyrat.opacity.models_type):
                    name = getattr(m, 'name', '')
                    if name == 'lecavelier':
                        store[f'{tag}_lec_pars'] = np.array(m.pars, float)
                    if name == 'deck':
                        store[f'{tag}_deck'] = np.array([m.pars[0], m.itop, m.rsurf, m.tsurf])
                    if mtype == 'cia':
                        key = 'cia_' + '_'.join(m.species)
                        if f'{key}_tab' not in store:
                            store[f'{key}_tab'] = m.tab_cross_section.astype(np.float64)
                            store[f'{key}_temps'] = m.temps
                            store[f'{key}_lohi'] = np.array([m._wn_lo_idx, m._wn_hi_idx])
                    if mtype == 'alkali':
                        store[f'{tag}_voigt_det'] = m.voigt_det(atm.temp)
                        store[f'{tag}_alk_cutoff'] = m.cutoff
                print(tag, 'W', spec.nwave, 'max rel dev from the reference golden',
                      float(np.max(np.abs(spec.spectrum / expected - 1))))
        np.savez_compressed(os.path.join(HERE, 'g9_reference_cases.npz'), **store)

        # ---- the golden vectors of tests/test_opacity_alkali.py and test_opacity_cia.py ----
        import pyratbay.atmosphere as pa
        import pyratbay.opacity as op
        import pyratbay.spectrum as ps
        g10 = {}
        pressure = pa.pressure('1e-8 bar', '1e2 bar', 6)
        g10['pressure'] = pressure
        for tag, cls, lo, hi in (('na', op.alkali.SodiumVdW, 0.55, 0.65),
                                 ('k', op.alkali.PotassiumVdW, 0.70, 0.84)):
            wn = ps.constant_resolution_spectrum(1e4 / hi, 1e4 / lo, 15000.0)
            model = cls(pressure, wn=wn, cutoff=1000.0)
            name = 'Na' if tag == 'na' else 'K'
            with np.load(f'{REF}/tests/expected/expected_alkali_{name}_opacity.npz') as d:
                cs1, cs2 = d['expected_cs1'], d['expected_cs2']
            np.testing.assert_allclose(model.calc_cross_section(np.tile(1000.0, 6)), cs1)
            np.testing.assert_allclose(model.calc_cross_section(np.tile(2500.0, 6)), cs2)
            g10[f'{tag}_wn'] = wn
            g10[f'{tag}_expected_cs1'] = cs1
            g10[f'{tag}_expected_cs2'] = cs2
        wn = ps.constant_resolution_spectrum(1e4 / 10.0, 1e4 / 0.5, 15.0)
        cia = op.Collision_Induced(
            f'{REF}/pyratbay/data/CIA/CIA_Borysow_H2H2_0060-7000K_0.6-500um.dat', wn=wn)
        with np.load(f'{REF}/tests/expected/expected_cia_H2H2_opacity.npz') as d:
            cs1, cs2, cs3 = d['expected_cs1'], d['expected_cs2'], d['expected_cs3']
        np.testing.assert_allclose(cia.calc_cross_section(np.tile(1200.0, 6)), cs1)
        np.testing.assert_allclose(cia.calc_cross_section(np.tile(3050.0, 6)), cs2)
        g10.update(cia_wn=wn, cia_tab=cia.tab_cross_section, cia_temps=cia.temps,
                   cia_lohi=np.array([cia._wn_lo_idx, cia._wn_hi_idx]),
                   cia_expected_cs1=cs1, cia_expected_cs2=cs2, cia_expected_cs3=cs3)
        np.savez_compressed(os.path.join(HERE, 'g10_opacity_goldens.npz'), **g10)
        print('g10_opacity_goldens.npz',
              os.path.getsize(os.path.join(HERE, 'g10_opacity_goldens.npz')) // 1024, 'KiB')
    finally:
        os.chdir(HERE)
        shutil.rmtree(work, ignore_errors=True)
    print('g9_reference_cases.npz',
          os.path.getsize(os.path.join(HERE, 'g9_reference_cases.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
