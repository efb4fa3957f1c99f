"""One-off sweep: every function of the drop-in modules pyratbay_amd.lib.* against the UNMODIFIED
reference extension compiled into oracle/_ref (same positional arguments: NumPy arrays with
random shapes, non-contiguous views, int64 index arrays read as 32-bit like ind.h does).
usage: python tools/fuzz_dropin.py [count]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RTOL = 1e-12


def view(rng, a):
    """the array itself, or an equal non-contiguous view of a padded copy"""
    a = np.asarray(a)
    if a.ndim == 0 or rng.random() < 0.5:
        return a.copy()
    pad = np.zeros(tuple(2 * n for n in a.shape), a.dtype)
    sl = tuple(slice(None, None, 2) for _ in a.shape)
    pad[sl] = a
    return pad[sl]


def close(a, b, what, rtol=RTOL, atol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind in 'iu':
        assert np.array_equal(a, b), what
    else:
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def one(rng, lib, ref):
    L = int(rng.integers(2, 40))
    W = int(rng.choice([1, 2, 63, 300]))
    radius = np.sort(rng.uniform(7e9, 8e9, L))[::-1].copy()
    h = -np.diff(radius)
    ec = 10.0**rng.uniform(-14, -8, (L, W)) * np.logspace(-3, 1, L)[:, None]
    temp = rng.uniform(300, 3000, L)
    wn = np.sort(rng.uniform(500, 20000, W))
    T, Tr = lib['_trapezoid'], ref.module('_trapezoid')
    # trapezoid / trapezoid2D / cumulative_sum
    y = rng.uniform(0, 1, L)
    close(T.trapezoid(view(rng, y), view(rng, h)), Tr.trapezoid(y, h), 'trapezoid')
    nint = rng.integers(0, L, W)                              # int64 like the package passes
    close(T.trapezoid2D(view(rng, ec), view(rng, h), view(rng, nint)),
          Tr.trapezoid2D(ec, h, nint.astype(np.intc)), 'trapezoid2D')
    out_a, out_b = np.zeros(L), np.zeros(L)
    thr = float(rng.choice([1e-30, 0.5 * np.sum(0.5 * h * (y[1:] + y[:-1])), 1e30]))
    ia, ib = T.cumulative_sum(out_a, y, h, thr), Tr.cumulative_sum(out_b, y, h, thr)
    assert ia == ib, 'cumulative_sum index'
    close(out_a[:ia + 1], out_b[:ib + 1], 'cumulative_sum')
    # plane-parallel depth (+ intensity on it)
    itop = int(rng.integers(0, max(1, L // 3)))
    ibottom = L if rng.random() < 0.6 else int(rng.integers(itop + 1, L + 1))
    maxdepth = float(rng.choice([0.3, 10.0, np.inf]))
    da, db = np.zeros((L, W)), np.zeros((L, W))
    ida, idb = np.full(W, L - 1, np.int64 if rng.random() < 0.5 else np.intc), np.full(W, L - 1, np.intc)
    T.plane_parallel_optical_depth(da, ida, view(rng, ec), view(rng, h), maxdepth, itop, ibottom)
    Tr.plane_parallel_optical_depth(db, idb, ec, h, maxdepth, itop, ibottom)
    close(da, db, 'plane depth')
    assert np.array_equal(np.asarray(ida).astype(np.int64) & 0xffffffff, idb.astype(np.int64) & 0xffffffff), 'plane ideep'
    B, Br = lib['_blackbody'], ref.module('_blackbody')
    last = idb.copy()
    Ba = B.blackbody_wn_2D(view(rng, wn), view(rng, temp))
    Bb = Br.blackbody_wn_2D(wn, temp)
    close(Ba, Bb, 'blackbody_wn_2D')
    Bc, Bd = np.zeros((L, W)), np.zeros((L, W))
    B.blackbody_wn_2D(wn, temp, Bc, last)
    Br.blackbody_wn_2D(wn, temp, Bd, last)
    close(Bc, Bd, 'blackbody_wn_2D last')
    t0 = float(temp[0])
    close(B.blackbody_wn(wn, t0), Br.blackbody_wn(wn, t0), 'blackbody_wn')
    mu = np.cos(np.radians(rng.uniform(0, 85, int(rng.integers(1, 6)))))
    rtop = itop
    close(T.intensity(view(rng, db), idb, view(rng, Bb), mu, rtop),
          Tr.intensity(db, idb, Bb, mu, rtop), 'intensity', 1e-10,
          1e-13 * float(np.max(Bb)))                        # differences of exponentials cancel
    # the transit loop of optic_depth.py:103-112 through optdepth
    from oracle import oracle as orc
    path = orc.transit_path(radius, itop)
    ia_, ib_ = np.full(W, -1, np.intc), np.full(W, -1, np.intc)
    for r in range(itop, min(ibottom, L)):
        ta = T.optdepth(ec[itop:r + 1], path[r], maxdepth, ia_, r)
        tb = Tr.optdepth(ec[itop:r + 1], path[r], maxdepth, ib_, r)
        close(ta, tb, f'optdepth row {r}')
        assert np.array_equal(ia_, ib_), 'optdepth ideep'
    # Simpson
    S, Sr = lib['_simpson'], ref.module('_simpson')
    n = int(rng.integers(2, 30))
    hs = rng.uniform(0.1, 2.0, n - 1)
    ga, gb = S.geth(hs), Sr.geth(hs)
    for x, y_ in zip(ga, gb):
        close(x, y_, 'geth')
    ys = rng.uniform(0, 1, (n, W))
    ni = rng.integers(0, n + 1, W)
    close(S.simps2D(view(rng, ys), hs, view(rng, ni), *gb), Sr.simps2D(ys, hs, ni.astype(np.intc), *gb), 'simps2D')
    if n % 2 == 1:
        close(S.simps(ys[:, 0].copy(), hs, *gb), Sr.simps(ys[:, 0].copy(), hs, *gb), 'simps')
    # cutils / _indices
    Cu, Cr = lib['cutils'], ref.module('cutils')
    close(Cu.ediff(view(rng, radius)), Cr.ediff(radius), 'ediff')
    grid = np.sort(rng.uniform(0, 100, int(rng.integers(2, 50))))
    # (values at or above the last element make the reference read one past the array's end:
    # binsearchapprox is called with hi = n)
    vals = rng.uniform(-10, grid[-1] * 0.999, int(rng.integers(1, 40)))
    close(Cu.arrbinsearch(vals, grid), Cr.arrbinsearch(vals, grid), 'arrbinsearch')
    I, Ir = lib['_indices'], ref.module('_indices')
    flags = (rng.random(int(rng.integers(1, 30))) < rng.choice([0.0, 0.3, 1.0])).astype(np.intc)
    for d in (-1, 7):
        assert I.ifirst(flags, d) == Ir.ifirst(flags, d) and I.ilast(flags, d) == Ir.ilast(flags, d), 'ifirst/ilast'
    # interp_ec / interp_ec_per_mol
    E, Er = lib['_extcoeff'], ref.module('_extcoeff')
    nmol, ntemp = int(rng.integers(1, 5)), int(rng.integers(2, 8))
    ttable = np.sort(rng.uniform(200, 3000, ntemp))
    etable = 10.0**rng.uniform(-30, -20, (nmol, ntemp, L, W))
    tl = rng.uniform(ttable[0], ttable[-1], L)
    tl[0] = ttable[rng.integers(0, ntemp)]
    dens = 10.0**rng.uniform(8, 18, (L, nmol))
    l1 = int(rng.integers(0, L))
    l2 = int(rng.integers(l1, L + 3))
    xa, xb = np.zeros((L, W)), np.zeros((L, W))
    E.interp_ec(xa, view(rng, etable), ttable, view(rng, tl), view(rng, dens), l1, l2)
    Er.interp_ec(xb, etable, ttable, tl, dens, l1, l2)
    close(xa, xb, 'interp_ec')
    pa, pb_ = np.zeros((nmol, L, W)), np.zeros((nmol, L, W))
    E.interp_ec_per_mol(pa, etable, ttable, tl, dens, l1, l2)
    Er.interp_ec_per_mol(pb_, etable, ttable, tl, dens, l1, l2)
    close(pa, pb_, 'interp_ec_per_mol')


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import ref
    if not ref.available():
        sys.exit('oracle/_ref is not built')
    from pyratbay_amd import engine
    engine.require_gpu()
    import importlib
    lib = {n: importlib.import_module('pyratbay_amd.lib.' + n)
           for n in ('_trapezoid', '_blackbody', '_simpson', 'cutils', '_indices', '_extcoeff')}
    bad = []
    for seed in range(count):
        try:
            one(np.random.default_rng(4000 + seed), lib, ref)
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
