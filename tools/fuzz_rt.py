"""One-off sweep of the remaining radiative-transfer kernels: two-stream (well-conditioned random
columns: increments from 1e-3 to 16, with / without internal flux and irradiation, rtop 0..3),
the patchy-cloud mix in transit geometry and the opaque deck, against the oracle.
usage: python tools/fuzz_rt.py [count]"""
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
    L = int(rng.integers(2, 60))
    W = int(rng.choice([1, 63, 700, 5000]))
    wn = np.sort(rng.uniform(300.0, 9000.0, W))
    temp = np.linspace(900.0, 2100.0, L) + rng.normal(0, 15.0, L)
    dtau = 10**rng.uniform(-3, 1.2, (L - 1, W))
    depth = np.vstack([np.zeros(W), np.cumsum(dtau, axis=0)])
    f_int = 10**rng.uniform(0, 2, W) if rng.random() < 0.7 else None
    top = 10**rng.uniform(2, 4, W) if rng.random() < 0.7 else None
    rtop = int(rng.integers(0, min(3, L - 1) + 1)) if rng.random() < 0.3 else 0
    want = orc.two_stream(depth, wn, temp, np.zeros(W) if f_int is None else f_int, top, rtop)
    down, up = eng.two_stream(eng.dev(depth), eng.dev(wn), eng.dev(temp),
                              None if f_int is None else eng.dev(f_int),
                              None if top is None else eng.dev(top), rtop)
    np.testing.assert_allclose(host(down), want[0], rtol=1e-10, atol=1e-10 * np.max(np.abs(want[0])))
    np.testing.assert_allclose(host(up), want[1], rtol=1e-10, atol=1e-10 * np.max(np.abs(want[1])))
    # patchy transit with an optional deck
    c = cases.column_case(seed=int(rng.integers(0, 10**6)), nlayers=L, nwave=W)
    itop = int(rng.integers(0, max(1, L // 3)))
    radius, ec = c['radius'], c['ec']
    ec_cloud = ec * 10.0**rng.uniform(-2, 1)
    fpatchy = float(rng.uniform(0, 1))
    deck_itop = int(rng.integers(itop + 1, L)) if rng.random() < 0.5 and L - itop > 2 else None
    rsurf = None
    if deck_itop is not None:
        f = float(rng.uniform(0.05, 0.95))
        rsurf = radius[deck_itop - 1] + f * (radius[deck_itop] - radius[deck_itop - 1])
    path = eng.dev(eng.pack_raypath(eng.transit_path(radius, itop), itop))
    spec, clear, cloudy = eng.patchy_transit_spectrum(eng.dev(ec), eng.dev(ec_cloud), fpatchy, path,
                                                      eng.dev(radius), c['rstar'], itop, 10.0,
                                                      rsurf, deck_itop)
    dcl, icl = orc.optical_depth_transit(ec, radius, itop, L, 10.0)
    want_clear = orc.transmission(dcl, radius, c['rstar'], icl, itop)
    ecc = ec.copy()
    ecc[itop:] += ec_cloud[itop:]
    ibottom = L if deck_itop is None else deck_itop + 1
    dcd, icd = orc.optical_depth_transit(ecc, radius, itop, ibottom, 10.0)
    want_cloudy = orc.transmission_deck(dcd, radius, c['rstar'], icd, itop, rsurf, deck_itop)
    np.testing.assert_allclose(host(clear), want_clear, rtol=1e-11)
    np.testing.assert_allclose(host(cloudy), want_cloudy, rtol=1e-11)
    np.testing.assert_allclose(host(spec), fpatchy * want_cloudy + (1 - fpatchy) * want_clear,
                               rtol=1e-11)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import oracle
    oracle.lib()
    from pyratbay_amd import engine
    engine.require_gpu()
    bad = []
    for seed in range(count):
        try:
            one(engine, oracle, np.random.default_rng(50000 + seed))
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
