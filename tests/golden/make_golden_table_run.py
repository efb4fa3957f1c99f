#!/usr/bin/env python3
"""Generate tests/golden/g8_table_run_{transit,emission}.npz: the REAL reference package run
on a sampled cross-section table plus continuum models plus band passes (build container
only; the package is imported as in make_golden_e2e.py):

    python tests/golden/make_golden_table_run.py

Steps, all inside a scratch directory: TLI from the bundled mock HITRAN H2O list ->
`runmode = opacity` (cross-section table on a temperature grid) -> `runmode = spectrum`
with that table, Borysow H2-H2 / H2-He CIA, a Lecavelier haze, H2 Rayleigh, the potassium
doublet and four top-hat band passes, for the transit and emission geometries.

Stored: the table, the atmosphere arrays handed to the opacity models, the total
extinction coefficient, optical depth, spectrum and band-integrated values, and the
band-pass sampling (indices, response, wavelengths)."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e          # noqa: E402

CFG_OPACITY = '''
[pyrat]
runmode = opacity
logfile = {work}/table.log
atmfile = {ref}/tests/inputs/atmosphere_uniform_test.atm
tlifile = {work}/mock_h2o.tli
sampled_cross_sec = {work}/table.npz
wl_low = 1.00 um
wl_high = 1.01 um
wnstep = 0.5
wnosamp = 120
voigt_extent = 60.0
voigt_cutoff = 8.0
nlor = 24
ndop = 12
tmin = 600
tmax = 2400
tstep = 300
ncpu = 1
verb = 0
'''

CFG_SPEC = '''
[pyrat]
runmode = spectrum
logfile = {work}/spec_{rt}.log
rt_path = {rt}
atmfile = {ref}/tests/inputs/atmosphere_uniform_test.atm
sampled_cross_sec = {work}/table.npz
continuum_cross_sec =
    {ref}/pyratbay/data/CIA/CIA_Borysow_H2H2_0060-7000K_0.6-500um.dat
    {ref}/pyratbay/data/CIA/CIA_Borysow_H2He_0050-3000K_0.3-030um.dat
rayleigh = rayleigh_H2
clouds = lecavelier 0.5 -3.2
alkali = potassium_vdw
obsfile = {work}/bands.dat
wl_low = 1.00 um
wl_high = 1.01 um
wnstep = 0.5
wnosamp = 120
rstar = 1.27 rsun
tstar = 5800.0
smaxis = 0.045 au
mplanet = 0.6 mjup
rplanet = 1.0 rjup
refpressure = 0.1 bar
radmodel = hydro_m
maxdepth = 10.0
ncpu = 1
verb = 0
'''

BANDS = '''# wl  half_width  name
@DATA
1.0015   0.0012   band_a
1.0040   0.0010   band_b
1.0065   0.0011   band_c
1.0088   0.0009   band_d
'''


def main():
    work = tempfile.mkdtemp(prefix='pb_table_')
    try:
        pb = e2e.reference_package(work)

        def run(cfg_text, name, **kw):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(cfg_text.format(work=work, ref=e2e.REF, **kw))
            return pb.run(cfg)

        run(e2e.CFG_TLI, 'tli')
        run(CFG_OPACITY, 'opacity')
        with open(os.path.join(work, 'bands.dat'), 'w') as f:
            f.write(BANDS)
        with np.load(os.path.join(work, 'table.npz'), allow_pickle=True) as f:
            table = {k: f[k] for k in f.files}
        table['species'] = np.array([str(s) for s in table['species']])
        for rt in ('transit', 'emission'):
            pyrat = run(CFG_SPEC, f'spec_{rt}', rt=rt)
            spec, atm, od = pyrat.spec, pyrat.atm, pyrat.od
            bandflux = pyrat.band_integrate()
            models = {name: model for name, model in
                      zip(pyrat.opacity.models_type, pyrat.opacity.models)}
            print(rt, 'models:', pyrat.opacity.models_type,
                  [getattr(m, 'name', '?') for m in pyrat.opacity.models])
            store = dict(
                rt_path=rt, wn=spec.wn,
                table_species=table['species'], table_temperature=table['temperature'],
                table_pressure=table['pressure'], table_wn=table['wavenumber'],
                table_opacity=table['opacity'],
                press=atm.press, temp=atm.temp, dens=atm.d, radius=atm.radius,
                species=np.array(atm.species), rtop=atm.rtop, rstar=atm.rstar,
                maxdepth=od.maxdepth, ec=np.array(pyrat.opacity.ec, float),
                depth=od.depth, ideep=od.ideep, spectrum=spec.spectrum, bandflux=bandflux,
                nbands=len(pyrat.obs.filters),
            )
            for i, band in enumerate(pyrat.obs.filters):
                store[f'band{i}_idx'] = np.asarray(band.idx)
                store[f'band{i}_response'] = band.response
                store[f'band{i}_wn'] = band.wn
                store[f'band{i}_wl'] = band.wl
                store[f'band{i}_height'] = band.height
                store[f'band{i}_photon'] = band.counting_type != 'energy'
            ncia = 0
            for m, mtype in zip(pyrat.opacity.models, pyrat.opacity.models_type):
                name = getattr(m, 'name', '')
                if mtype == 'cia':
                    store[f'cia{ncia}_tab'] = m.tab_cross_section
                    store[f'cia{ncia}_temps'] = m.temps
                    store[f'cia{ncia}_lohi'] = np.array([m._wn_lo_idx, m._wn_hi_idx])
                    store[f'cia{ncia}_species'] = np.array(m.species)
                    ncia += 1
                if mtype == 'alkali':
                    store['alk_voigt_det'] = m.voigt_det(atm.temp)
                if name == 'lecavelier':
                    store['lec_pars'] = np.array(m.pars, float)
                if name == 'potassium_vdw':
                    store['alk_cutoff'] = m.cutoff
            if rt == 'emission':
                store.update(quadrature_mu=spec.quadrature_mu,
                             quadrature_weights=np.ravel(spec.quadrature_weights))
            np.savez_compressed(os.path.join(HERE, f'g8_table_run_{rt}.npz'), **store)
            print(rt, 'W', spec.nwave, 'L', atm.nlayers, 'bandflux', bandflux)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for f in sorted(os.listdir(HERE)):
        if f.startswith('g8_'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
