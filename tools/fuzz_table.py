"""One-off sweep of the sampled-cross-section path: TableSpectrum.eval (one walker) against the
oracle chain, eval_bands (walker batch, transit and emission geometry, shared or per-walker radius,
random chunk sizes, rejected walkers) against per-walker eval, pb_loglike and the band integration
against NumPy.  usage: python tools/fuzz_table.py [count]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def host(t):
    return t.cpu().numpy()


def one(eng, orc, rng):
    import cases
    from pyratbay_amd import synth
    nspec = int(rng.integers(1, 7))
    ntemp = int(rng.integers(2, 12))
    L = int(rng.integers(2, 45))
    W = int(rng.choice([2, 63, 257, 1500, 3001]))
    rt = str(rng.choice(['transit', 'emission']))
    itop = int(rng.integers(0, max(1, L // 4)))
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    wn = g['wn']
    ttable = np.sort(rng.uniform(200.0, 3000.0, ntemp))
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * 10.0**rng.uniform(-1, 1, (nspec, 1, 1, W))
    c = cases.column_case(seed=int(rng.integers(0, 10**6)), nlayers=L, nwave=W)
    radius0 = c['radius']
    rstar = c['rstar']
    model = eng.TableSpectrum(etable, ttable, wn, radius0, rstar, rt_path=rt, itop=itop)
    nb = int(rng.integers(1, 6))
    bands = []
    for _ in range(nb):
        lo = int(rng.integers(0, W - 1))
        hi = int(rng.integers(lo + 2, W + 1)) if W - lo >= 2 else W
        resp = rng.uniform(0.1, 1.0, hi - lo)
        bands.append((lo, resp, float(rng.uniform(0.5, 2.0))))
    pb = eng.PassBands(wn, bands)
    nw = int(rng.choice([1, 2, 9, 70]))
    tmid = 0.5 * (ttable[0] + ttable[-1])
    temps = np.clip(tmid * (1 + 0.3 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.9, 1.1, L),
                    ttable[0], ttable[-1])
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    per_walker_radius = rng.random() < 0.5
    radius = np.array([np.sort(radius0 * (1 + 0.01 * rng.uniform(-1, 1)))[::-1] for _ in range(nw)]) \
        if per_walker_radius else None
    bad = []
    if nw > 2 and rng.random() < 0.5:
        bad = [int(rng.integers(0, nw))]
        temps[bad[0], int(rng.integers(0, L))] = ttable[-1] + 1.0
    chunk = int(rng.choice([1, 3, 64]))
    got = host(model.eval_bands(eng.dev(temps), eng.dev(dens), pb,
                                radius=None if radius is None else eng.dev(radius), chunk=chunk))
    assert got.shape == (nw, nb)
    for w in bad:
        assert np.all(np.isinf(got[w])) and np.all(got[w] > 0)
    ok = [w for w in range(nw) if w not in bad]
    if not np.all(np.isfinite(got[ok])):
        raise AssertionError(f'non-finite band flux: rt={rt} nspec={nspec} ntemp={ntemp} L={L} W={W} itop={itop} nw={nw} '
                             f'chunk={chunk} per_walker_radius={per_walker_radius} bad={bad} bands={[(b[0], len(b[1])) for b in bands]} '
                             f'got={got[ok][:2]}')
    for w in list(rng.choice(ok, min(len(ok), 3), replace=False)):
        rad = radius0 if radius is None else radius[w]
        model.set_radius(rad)
        spec = host(model.eval(temps[w], dens[w]))
        # oracle chain for the one-walker spectrum
        ec = np.zeros((L, W))
        orc.interp_ec(ec, etable, ttable, temps[w], dens[w], 0, L)
        if rt == 'transit':
            depth, ideep = orc.optical_depth_transit(ec, rad, itop, L, model.maxdepth)
            want = orc.transmission(depth, rad, rstar, ideep, itop)
        else:
            depth = np.zeros((L, W))
            ideep = np.full(W, L - 1, np.int32)
            orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(rad), model.maxdepth,
                                             itop, L)
            inten = orc.intensity(depth, ideep, orc.blackbody_wn_2D(wn, temps[w]),
                                  host(model.mu), itop)
            want = np.sum(inten * host(model.weights)[:, None], axis=0)
        np.testing.assert_allclose(spec, want, rtol=1e-10, atol=1e-13 * np.max(np.abs(want)))
        flux = [np.trapezoid(want[s:s + len(r)] * r, wn[s:s + len(r)]) * h for s, r, h in bands]
        np.testing.assert_allclose(got[w], flux, rtol=1e-9, atol=1e-12 * np.max(np.abs(flux)))
    # log-likelihood of the batch
    data = rng.uniform(0.5, 1.5, nb) * np.nanmean(got[ok], axis=0)
    unc = rng.uniform(0.01, 0.1, nb) * np.abs(data)
    ll = host(eng.loglike(eng.dev(got), eng.dev(data), eng.dev(unc)))
    for w in range(nw):
        if w in bad:
            assert ll[w] == -1.0e98
        else:
            want_ll = (-0.5 * np.sum(((data - got[w]) / unc)**2.0)
                       - 0.5 * np.sum(np.log(2.0 * np.pi * unc**2.0)))   # retrieval_tools.py:98-101
            np.testing.assert_allclose(ll[w], want_ll, rtol=1e-12, atol=1e-12)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import oracle
    oracle.lib()
    from pyratbay_amd import engine
    engine.require_gpu()
    bad = []
    for seed in range(count):
        try:
            one(engine, oracle, np.random.default_rng(30000 + seed))
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
