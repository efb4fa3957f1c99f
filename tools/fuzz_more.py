"""One-off randomized sweeps beyond the committed tests: co-add grouping on the device against the
host pass (random densities, duplicates, lines beyond the grid, empty isotopes) and the walker
batch kernels (interpolation, transit incl. the two-column retrieval form) against the oracle.
usage: python tools/fuzz_more.py [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def host(t):
    return t.cpu().numpy()


def grouping(eng, rng):
    from pyratbay_amd import synth
    niso = int(rng.integers(1, 5))
    nwave = int(rng.integers(2, 4000))
    osamp = int(rng.choice([1, 2, 6, 24, 180]))
    g = synth.spectral_grid(4000.0, 4000.0 + (nwave - 1) * 0.05 + 0.01, 0.05, osamp)
    lo, hi = g['own'][0], g['own'][-1]
    counts = rng.integers(0, int(rng.choice([3, 50, 3000, 40000])), niso)
    if rng.random() < 0.2:
        counts[rng.integers(0, niso)] = 0
    lwn, lid = [], []
    for i, c in enumerate(counts):
        span = rng.choice([1.0, 1.2, 0.05])            # inside, beyond both ends, crowded
        w = rng.uniform(lo - (span - 1) * 10, lo + (hi - lo) * span, c) if span != 0.05 \
            else rng.uniform(lo, lo + (hi - lo) * 0.05, c)
        if c and rng.random() < 0.5:                   # exact duplicates and exact grid hits
            w[rng.integers(0, c, max(1, c // 7))] = w[rng.integers(0, c)]
            w[rng.integers(0, c, max(1, c // 9))] = g['own'][rng.integers(0, len(g['own']), max(1, c // 9))]
        lwn.append(np.sort(w))
        lid.append(np.full(c, i))
    lwn, lid = np.concatenate(lwn), np.concatenate(lid).astype(np.int32)
    n = len(lwn)
    elow, gf = rng.uniform(0, 8000, n), 10**rng.uniform(-12, -6, n)
    os.environ['PB_LINES_HOST'] = '1'
    ref = eng.LineList(lwn, elow, gf, lid, niso, g['own'])
    os.environ.pop('PB_LINES_HOST')
    ll = eng.LineList(lwn, elow, gf, lid, niso, g['own'])
    assert (ll.ninrange, ll.ngroups, ll.nadd) == (ref.ninrange, ref.ngroups, ref.nadd), \
        (ll.ninrange, ll.ngroups, ll.nadd, ref.ninrange, ref.ngroups, ref.nadd)
    for a, b in zip(ll.groups(), ref.groups()):
        assert np.array_equal(a, b)
    ref.close()
    ll.close()


def batch(eng, orc, rng):
    import cases
    nmol = int(rng.integers(1, 9))
    ntemp = int(rng.integers(2, 12))
    L = int(rng.integers(2, 60))
    W = int(rng.choice([1, 63, 257, 1500]))
    nw = int(rng.choice([1, 2, 17, 70]))
    ttable = np.sort(rng.uniform(200.0, 3000.0, ntemp))
    etable = 10.0**rng.uniform(-30, -20, (nmol, ntemp, L, W))
    temps = rng.uniform(ttable[0], ttable[-1], (nw, L))
    temps[0, :] = ttable[rng.integers(0, ntemp, L)]
    dens = 10.0**rng.uniform(8, 18, (nw, L, nmol))
    got = host(eng.interp_ec_batch(eng.dev(etable), eng.dev(ttable), eng.dev(temps), eng.dev(dens)))
    for w in rng.choice(nw, min(nw, 3), replace=False):
        want = np.zeros((L, W))
        orc.interp_ec(want, etable, ttable, temps[w], dens[w], 0, L)
        np.testing.assert_allclose(got[w], want, rtol=1e-12)
    # transit batch on a physical column problem
    c = cases.column_case(seed=int(rng.integers(0, 10**6)), nlayers=L, nwave=W)
    itop = int(rng.integers(0, max(1, L // 3)))
    ibottom = L if rng.random() < 0.6 else int(rng.integers(itop + 1, L + 1))
    maxdepth = float(rng.choice([0.3, 10.0, np.inf]))
    ecs = np.array([c['ec'] * 10.0**rng.uniform(-1, 1) for _ in range(nw)])
    radius = np.array([np.sort(c['radius'] * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    rad_d = eng.dev(radius)
    path = eng.transit_path_device(rad_d, itop)
    rows = rng.choice(['', '8', '16'])
    if rows:
        os.environ['PB_TRANSIT_ROWS'] = rows
    spec, depth, ideep = eng.transit_spectrum_batch(eng.dev(ecs), path, rad_d, c['rstar'], itop,
                                                    ibottom, maxdepth, want_depth=True)
    only = eng.transit_spectrum_batch(eng.dev(ecs), path, rad_d, c['rstar'], itop, ibottom,
                                      maxdepth)
    os.environ.pop('PB_TRANSIT_ROWS', None)
    for w in rng.choice(nw, min(nw, 3), replace=False):
        # the oracle on the DEVICE's ray paths (the host form squares with pow)
        wd, wi = orc.optical_depth_transit(ecs[w], radius[w], itop, ibottom, maxdepth)
        ws = orc.transmission(wd, radius[w], c['rstar'], wi, itop)
        assert np.array_equal(host(ideep[w]), wi), 'ideep'
        np.testing.assert_allclose(host(depth[w]), wd, rtol=1e-10, atol=0)
        np.testing.assert_allclose(host(spec[w]), ws, rtol=1e-10)
        np.testing.assert_allclose(host(only[w]), ws, rtol=1e-10)


def voigt(eng, orc, rng):
    """Random width grids: log-spaced like voigt.py builds them (with dlratio aliases) and raw
    random widths with explicit alias columns, random fine step and phase count."""
    from pyratbay_amd import synth
    if rng.random() < 0.5:
        case = synth.lbl_case(int(rng.integers(200, 3000)), int(rng.integers(2, 12)), 10,
                              wnstep=float(rng.choice([0.01, 0.05, 0.25, 1.0])),
                              nlor=int(rng.integers(2, 20)), ndop=int(rng.integers(2, 10)),
                              extent=float(rng.uniform(5, 120)), cutoff=float(rng.uniform(1, 30)),
                              dlratio=float(rng.choice([0.02, 0.1, 0.5])),
                              ptop=10.0**rng.uniform(-8, -3), pbottom=10.0**rng.uniform(-1, 2))
        vg, g = case['voigt'], case['grid']
        lor, dop, size_in, dwn, osamp = vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], g['wnosamp']
    else:
        nlor, ndop = int(rng.integers(1, 8)), int(rng.integers(1, 6))
        dwn = 10.0**rng.uniform(-4, -1.5)
        lor = np.sort(10.0**rng.uniform(-5, 0, nlor))
        dop = np.sort(10.0**rng.uniform(-3.5, -0.5, ndop))
        size_in = rng.integers(1, 400, (nlor, ndop))
        if rng.random() < 0.3:                             # one huge cell: the QUICK regime
            size_in[rng.integers(0, nlor), rng.integers(0, ndop)] = 50000 + int(rng.integers(0, 300))
        alias = rng.random((nlor, ndop)) < 0.25
        alias[:, 0] = False
        size_in = np.where(alias, 0, size_in)
        osamp = int(rng.choice([1, 2, 5, 12, 24]))
    size = np.array(size_in).copy()
    index = np.zeros_like(size)
    profile = np.zeros(int(np.sum(2 * np.maximum(size, 1) + 1)) + 8)
    orc.voigt_grid(profile, size, index, lor, dop, dwn)
    vt = eng.VoigtTable.build(lor, dop, size_in, dwn, osamp)
    assert np.array_equal(vt.size, size) and np.array_equal(vt.index, index)
    np.testing.assert_allclose(vt.flat(), profile[:vt.nprofile], rtol=2e-12)
    back = eng.VoigtTable.from_flat(profile[:vt.nprofile].copy(), size, index, lor, dop, osamp)
    assert np.array_equal(back.flat(), profile[:vt.nprofile])
    vt.close()
    back.close()


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    import traceback
    from oracle import oracle
    oracle.lib()
    from pyratbay_amd import engine
    engine.require_gpu()
    bad = []
    for seed in range(count):
        for name, fn in (('grouping', lambda r: grouping(engine, r)),
                         ('batch', lambda r: batch(engine, oracle, r)),
                         ('voigt', lambda r: voigt(engine, oracle, r))):
            try:
                fn(np.random.default_rng(9000 + seed))
            except Exception:                              # noqa: BLE001
                bad.append((name, seed))
                print(f'FAIL {name} seed {seed}')
                traceback.print_exc(limit=4)
        if seed % 20 == 19:
            print(f'{seed + 1} seeds, {len(bad)} failures', flush=True)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
