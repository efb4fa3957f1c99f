#!/usr/bin/env python3
"""Fixture G19: what the reference makes of an emission spectrum AFTER the radiative transfer --
the dilution factor, the planet-to-star flux ratio of the eclipse geometry, the f_lambda unit
conversion of eval(), and the band fluxes of emission / eclipse runs with filters
(pyrat/spectrum.py:394-405, pyrat/pyrat_obj.py:323-329, 649-668).  Build container only:

    python tests/golden/make_golden_observables.py

Runs of the real package (imported as in make_golden_e2e.py) from the reference's own
spectrum_{emission,eclipse}[_filters]_test.cfg without `sampled_cross_sec` (that table needs a
HITRAN download): per case the plane-parallel flux the run produced BEFORE the scalings (taken
from the same configuration run without f_dilution as rt_path = emission), and everything the
scalings read and write.  Only data is stored."""
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
REMOVE = ['sampled_cross_sec']


def bands_of(pyrat, store, tag):
    """The pass bands as the package holds them after set_sampling: index subset, response,
    height (PassBand.integrate reads exactly these, spec_tools.py:193-233)."""
    nb = len(pyrat.obs.filters)
    store[f'{tag}_nbands'] = np.array(nb)
    for b, band in enumerate(pyrat.obs.filters):
        store[f'{tag}_band{b}_idx'] = np.asarray(band.idx)
        store[f'{tag}_band{b}_response'] = np.asarray(band.response, float)
        store[f'{tag}_band{b}_height'] = np.array(band.height, float)
        store[f'{tag}_band{b}_counting'] = np.array(band.counting_type == 'photon')
        # (photon counting multiplies by band.wl: the wavelengths of the band's own samples)
        assert np.allclose(band.wl, 1e4 / pyrat.spec.wn[band.idx], rtol=1e-14)
        assert np.array_equal(band.wn, pyrat.spec.wn[band.idx])


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_obs_')
    store = {}
    try:
        pb = e2e.reference_package(work)
        import pyratbay.constants as pc
        os.chdir(work)
        store['um'] = np.array(pc.um)
        # ---- f_dilution on emission and eclipse runs (tests/test_emission.py:204-219,
        #      tests/test_eclipse.py:205-219: spectrum = 0.75 x the undiluted one)
        for rt in ('emission', 'eclipse'):
            base = f'{REF}/tests/configs/{rc.BASE[rt]}'
            plain = pb.run(rc.make_config(work, base, {}, REMOVE, f'{rt}_plain'))
            dil = pb.run(rc.make_config(work, base, {'f_dilution': '0.75'}, REMOVE, f'{rt}_dil'))
            flux = pb.run(rc.make_config(work, base, {'rt_path': 'emission'}, REMOVE,
                                         f'{rt}_flux'))
            store[f'{rt}_wn'] = plain.spec.wn
            store[f'{rt}_flux'] = np.copy(flux.spec.spectrum)        # before any scaling
            store[f'{rt}_plain_spectrum'] = np.copy(plain.spec.spectrum)
            store[f'{rt}_plain_fplanet'] = np.copy(plain.spec.fplanet)
            store[f'{rt}_dil_spectrum'] = np.copy(dil.spec.spectrum)
            store[f'{rt}_dil_fplanet'] = np.copy(dil.spec.fplanet)
            if rt == 'eclipse':
                store['eclipse_starflux'] = np.copy(plain.spec.starflux)
                store['eclipse_radii'] = np.array([plain.atm.rplanet, plain.atm.rstar])
                assert np.array_equal(plain.spec.eclipse, plain.spec.spectrum)
            np.testing.assert_allclose(dil.spec.spectrum, 0.75 * plain.spec.spectrum, rtol=1e-13)
        # ---- filters: eval() + band_integrate() (tests/test_emission.py:393-410,
        #      tests/test_eclipse.py:329-346)
        for rt in ('emission', 'eclipse'):
            base = f'{REF}/tests/configs/spectrum_{rt}_filters_test.cfg'
            pyrat = pb.run(rc.make_config(work, base, {}, REMOVE, f'{rt}_filters'))
            ev_spectrum, ev_bandflux = pyrat.eval(pyrat.ret.params, retmodel=True)
            bandflux = pyrat.band_integrate()
            np.testing.assert_allclose(ev_bandflux, bandflux, rtol=1e-13)
            tag = f'{rt}_filters'
            store[f'{tag}_wn'] = pyrat.spec.wn
            store[f'{tag}_spectrum'] = np.copy(pyrat.spec.spectrum)
            store[f'{tag}_fplanet'] = np.copy(pyrat.spec.fplanet)
            store[f'{tag}_bandflux'] = np.copy(bandflux)
            bands_of(pyrat, store, tag)
            if rt == 'eclipse':
                store[f'{tag}_starflux'] = np.copy(pyrat.spec.starflux)
                store[f'{tag}_bandflux_star'] = np.copy(pyrat.obs.bandflux_star)
                store[f'{tag}_radii'] = np.array([pyrat.atm.rplanet, pyrat.atm.rstar])
        # ---- f_lambda: eval() converts erg s-1 cm-2 cm to W m-2 um-1 (pyrat_obj.py:323-329)
        base = f'{REF}/tests/configs/spectrum_emission_filters_test.cfg'
        reset = {'rt_path': 'f_lambda', 'distance': '10.0 parsec'}
        pyrat = pb.run(rc.make_config(work, base, reset, REMOVE, 'f_lambda'))
        store['f_lambda_wn'] = pyrat.spec.wn
        store['f_lambda_run_spectrum'] = np.copy(pyrat.spec.spectrum)   # run(): still erg s-1 cm-2 cm
        ev_spectrum, ev_bandflux = pyrat.eval(pyrat.ret.params, retmodel=True)
        store['f_lambda_flux'] = np.copy(pyrat.spec.fplanet) if pyrat.spec.fplanet is not None \
            else np.zeros(0)
        store['f_lambda_spectrum'] = np.copy(ev_spectrum)
        store['f_lambda_bandflux'] = np.copy(ev_bandflux)
        store['f_lambda_scalars'] = np.array([pyrat.atm.rplanet, pyrat.atm.distance])
        bands_of(pyrat, store, 'f_lambda')
        for key in [k for k in store if k.startswith('f_lambda_band')]:
            # (same obsfile, same grid as the emission run: stored once)
            assert np.array_equal(store[key], store[key.replace('f_lambda', 'emission_filters')])
            del store[key]
        # arrays that are copies of one another are stored once: `same_as` names the kept one
        same, kept = {}, {}
        for k in list(store):
            v = store[k]
            if v.ndim != 1 or v.size < 100:
                continue
            sig = (v.shape, v.dtype.str, v.tobytes())
            if sig in kept:
                same[k] = kept[sig]
                del store[k]
            else:
                kept[sig] = k
        store['same_as'] = np.array([f'{k}={v}' for k, v in sorted(same.items())])
        np.savez_compressed(os.path.join(HERE, 'g19_observables.npz'), **store)
        for k, v in store.items():
            if v.ndim and v.size > 4:
                print(k, v.shape, v.dtype)
    finally:
        os.chdir(HERE)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
