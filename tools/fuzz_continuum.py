"""One-off sweep of the continuum kernels (pb_continuum, pb_alkali_cross_section) against the
oracle's restatement of the reference classes (oracle/continuum.py, pinned by fixture G7): random
grids (ascending wavenumbers over the alkali lines and the H- edge), pressures, temperatures,
densities, haze / gray-cloud parameters, random CIA tables.  usage: python tools/fuzz_continuum.py [n]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(eng, ct, cont, rng):
    L = int(rng.integers(2, 40))
    W = int(rng.choice([2, 63, 700, 4000]))
    lo = float(rng.choice([300.0, 3000.0, 9000.0]))
    wn = np.sort(rng.uniform(lo, lo + rng.choice([500.0, 9000.0, 25000.0]), W))
    wn = np.unique(wn)
    W = len(wn)
    pressure = np.logspace(rng.uniform(-8, -4), rng.uniform(-1, 2), L)
    temp = rng.uniform(200.0, 3000.0, L)
    dens_tot = cont.nominal_density(pressure, temp)
    d = {s: dens_tot * 10.0**rng.uniform(-8, 0) for s in ('H', 'He', 'H2', 'e-', 'Na', 'K')}

    def run(models):
        ec = eng.dev(np.zeros((L, W)))
        ct.Continuum(wn, pressure, models).add(ec, temp, d)
        return ec.cpu().numpy()

    def close(got, want, what, rtol=1e-11):
        assert np.array_equal(got == 0, want == 0), f'{what}: zero pattern'
        np.testing.assert_allclose(got, want, rtol=rtol, err_msg=what)

    total = np.zeros((L, W))
    models = []
    for sp in ('H', 'He', 'H2', 'e-'):
        want = cont.rayleigh_cross_section(wn, sp) * d[sp][:, None]
        close(run([ct.Kurucz(wn, sp)]), want, f'rayleigh {sp}')
        total += want
        models.append(ct.Kurucz(wn, sp))
    lec_pars = [float(rng.uniform(-2, 4)), float(rng.uniform(-6, 0))]
    lec = ct.Lecavelier(pressure, wn=wn)
    lec.calc_cross_section(lec_pars)
    want = cont.lecavelier_cross_section(wn, lec_pars) * dens_tot[:, None]
    close(run([lec]), want, 'lecavelier')
    total += want
    models.append(lec)
    gray = ct.CCSgray(pressure, wn)
    ptop, pbot = sorted(rng.uniform(-8, 2, 2))
    gray.pars[:] = [float(rng.uniform(-2, 4)), float(ptop), float(pbot)]
    want = (cont.gray_layer_cross_section(pressure, gray.pars) * dens_tot)[:, None] * np.ones(W)
    close(run([gray]), want, 'gray')
    total += want
    models.append(gray)
    # H-
    bf, ff = cont.hminus_cross_sections(wn, temp)
    want = (bf + ff) * (d['H'] * d['e-'])[:, None]
    close(run([ct.Hydrogen_Ion(wn)]), want, 'H-', 1e-10)
    total += want
    models.append(ct.Hydrogen_Ion(wn))
    # CIA: a random table on a sub-range of the grid
    if W >= 8:
        nt = int(rng.integers(2, 9))
        temps = np.sort(rng.uniform(50.0, 3500.0, nt))
        temps[0], temps[-1] = min(temps[0], temp.min()), max(temps[-1], temp.max())
        ilo = int(rng.integers(0, W // 2))
        ihi = int(rng.integers(ilo + 2, W + 1))
        tab = np.zeros((nt, W))                            # full grid width, zero outside [ilo, ihi)
        tab[:, ilo:ihi] = 10.0**rng.uniform(-50, -40, (nt, ihi - ilo))
        m = ct.Collision_Induced.__new__(ct.Collision_Induced)
        m.species, m.nspec = ['H2', 'He'], 2
        m.tab_cross_section, m.temps = tab, temps
        m.ntemp, m.tmin, m.tmax = nt, temps.min(), temps.max()
        m._wn_lo_idx, m._wn_hi_idx = ilo, ihi
        full = cont.cia_cross_section(tab, temps, temp, ilo, ihi)
        want = full * (d['H2'] * d['He'])[:, None]
        close(run([m]), want, 'cia', 1e-10)
        total += want
        models.append(m)
    # alkali doublets
    for cls, sp in ((ct.SodiumVdW, 'Na'), (ct.PotassiumVdW, 'K')):
        a = cls(pressure, wn=wn)
        vd = a.voigt_det(temp)
        cs = cont.alkali_cross_section(pressure * 1e6, wn, temp, vd, a.detuning, a.mass, a.lpar,
                                       a.Z, a.cutoff, np.array(a.wn0), np.array(a.gf))
        want = cs * d[sp][:, None]
        close(run([a]), want, f'alkali {sp}', 1e-10)
        total += want
        models.append(a)
    np.testing.assert_allclose(run(models), total, rtol=1e-10)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import continuum as cont
    from pyratbay_amd import engine, continuum as ct
    engine.require_gpu()
    bad = []
    for seed in range(count):
        try:
            one(engine, ct, cont, np.random.default_rng(70000 + seed))
        except Exception:                                  # noqa: BLE001
            bad.append(seed)
            print('FAIL seed', seed)
            traceback.print_exc(limit=3)
        if seed % 50 == 49:
            print(f'{seed + 1} seeds, {len(bad)} failures', flush=True)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
