#!/usr/bin/env python3
"""Generate tests/golden/g7_continuum.npz from the REAL reference package's continuum
opacity models (build container only; see make_golden_e2e.py for how the package is
imported from a scratch copy):

    python tests/golden/make_golden_continuum.py

Models exercised: opacity.rayleigh.Kurucz (H, He, H2, e-),
opacity.clouds.Lecavelier, opacity.clouds.CCSgray, opacity.Collision_Induced (Borysow H2-H2
and H2-He tables shipped with the package, spline-resampled to the model grid by
lib._spline), opacity.Hydrogen_Ion (John 1988), opacity.alkali.SodiumVdW / PotassiumVdW
(lib._alkali).  The fixture holds the inputs each model hands to its arithmetic and the
cross sections / extinction coefficients it returns.  Only data is written."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e          # noqa: E402


def main():
    work = tempfile.mkdtemp(prefix='pb_cont_')
    try:
        pb = e2e.reference_package(work)
        import pyratbay.opacity as op
        import pyratbay.constants as pc
        from pyratbay.lib import _spline as sp

        nlayers = 12
        wn = np.arange(2000.0, 26000.0, 24.0)              # 1000 samples, 0.385-5 um
        pressure = np.logspace(-6, 2, nlayers)             # bar
        temp = np.linspace(950.0, 2900.0, nlayers)
        temp[3] = 1000.0                                   # a CIA table node
        dens_tot = pressure * pc.bar / (pc.k * temp)
        vmr = dict(H2=0.85, He=0.148, H=1e-3, e=1e-7, Na=2e-6, K=1e-7)
        store = dict(wn=wn, pressure=pressure, temp=temp, dens_tot=dens_tot,
                     **{f'vmr_{k}': v for k, v in vmr.items()})

        # Rayleigh (rayleigh.py:13-107)
        for species, key in (('H', 'H'), ('He', 'He'), ('H2', 'H2'), ('e-', 'e')):
            model = op.rayleigh.Kurucz(wn, species)
            d = vmr[key] * dens_tot
            store[f'ray_{key}_cs'] = model.cross_section
            store[f'ray_{key}_ec'] = model.calc_extinction_coefficient(d)

        # Lecavelier (lecavelier.py:14-100) and constant-cross-section gray cloud
        lec = op.clouds.Lecavelier(pressure, wn=wn)
        store['lec_pars'] = np.array([0.7, -3.6])
        store['lec_ec'] = lec.calc_extinction_coefficient(temp, pars=store['lec_pars'])
        store['lec_cs'] = lec.cross_section
        gray = op.clouds.CCSgray(pressure, wn)
        store['gray_pars'] = np.array([1.3, -3.5, 0.5])
        store['gray_ec'] = np.array(gray.calc_extinction_coefficient(temp, pars=store['gray_pars']))

        # CIA (cia.py:20-215; lib/_spline second_deriv, splinterp_1D, lin_interp_2D)
        for tag, fname, d2 in (
                ('h2h2', 'CIA_Borysow_H2H2_0060-7000K_0.6-500um.dat', ('H2', 'H2')),
                ('h2he', 'CIA_Borysow_H2He_0050-7000K_0.5-031um.dat', ('H2', 'He'))):
            path = f'{pc.ROOT}/pyratbay/data/CIA/{fname}'
            cia = op.Collision_Induced(path, wn=wn)
            dens = np.array([vmr[d2[0]] * dens_tot, vmr[d2[1]] * dens_tot]).T
            store[f'cia_{tag}_tab'] = cia.tab_cross_section
            store[f'cia_{tag}_temps'] = cia.temps
            store[f'cia_{tag}_lohi'] = np.array([cia._wn_lo_idx, cia._wn_hi_idx])
            store[f'cia_{tag}_dens'] = dens
            store[f'cia_{tag}_cs'] = cia.calc_cross_section(temp).copy()
            store[f'cia_{tag}_ec'] = cia.calc_extinction_coefficient(temp, dens)
        # the raw H2-He table and its spline resampling, for the front-end's own reader
        absorption, species, temps, tab_wn = pb.io.read_cs(path)
        store['cia_raw_absorption'] = absorption[:, ::4][:3]
        store['cia_raw_wn'] = tab_wn[::4]
        y = store['cia_raw_absorption'][1]
        ddev = sp.second_deriv(y, store['cia_raw_wn'])
        store['cia_raw_ddev'] = ddev
        store['cia_raw_interp'] = sp.splinterp_1D(y, store['cia_raw_wn'], ddev, wn, 0.0)
        store['amagat'] = pc.amagat

        # H- (hydrogen_ion.py:17-276)
        hm = op.Hydrogen_Ion(wn)
        dens = np.array([vmr['H'] * dens_tot, vmr['e'] * dens_tot]).T
        store['hm_sigma_bf'] = hm.sigma_bf
        store['hm_cs_bf'] = hm.cross_section_bound_free(temp)
        store['hm_cs_ff'] = hm.cross_section_free_free(temp)
        store['hm_dens'] = dens
        store['hm_ec'] = hm.calc_extinction_coefficient(temp, dens)

        # alkali (alkali.py:28-391, src_c/_alkali.c:30-106)
        for tag, cls, key in (('na', op.alkali.SodiumVdW, 'Na'), ('k', op.alkali.PotassiumVdW, 'K')):
            model = cls(pressure, wn=wn, cutoff=4500.0)
            d = vmr[key] * dens_tot
            store[f'alk_{tag}_voigt_det'] = model.voigt_det(temp)
            store[f'alk_{tag}_scalars'] = np.array([model.detuning, model.mass, model.lpar,
                                                    model.Z, model.cutoff])
            store[f'alk_{tag}_wn0'] = np.array(model.wn0)
            store[f'alk_{tag}_gf'] = np.array(model.gf)
            store[f'alk_{tag}_dwave'] = np.array(model._dwave)
            store[f'alk_{tag}_cs'] = model.calc_cross_section(temp).copy()
            store[f'alk_{tag}_ec'] = model.calc_extinction_coefficient(temp, d)
        store['bar'] = pc.bar
        store['k_boltz'] = pc.k
        np.savez_compressed(os.path.join(HERE, 'g7_continuum.npz'), **store)
        print('g7_continuum.npz', os.path.getsize(os.path.join(HERE, 'g7_continuum.npz')) // 1024,
              'KiB;', len(store), 'arrays; W', len(wn), 'L', nlayers)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
