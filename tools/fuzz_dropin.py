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


def heavy(rng, lib, ref, seed):
    """vprofile.grid and the per-layer _extcoeff.extinction with HOST arrays in the package's own
    dtypes (int64 sizes / indices / divisors / isotope maps, read as 32-bit like ind.h)."""
    from pyratbay_amd import synth
    V, Vr = lib['vprofile'], ref.module('vprofile')
    E, Er = lib['_extcoeff'], ref.module('_extcoeff')
    niso = int(rng.integers(1, 4))
    osamp = int(rng.choice([6, 12, 24]))
    case = synth.lbl_case(int(rng.integers(3, 1500)), int(rng.integers(1, 4)),
                          int(rng.integers(1, 4000)), wnstep=float(rng.choice([0.05, 0.2])),
                          # (a width grid much coarser than this lets a layer's nearest profile be
                          # narrower than its dynamic-sampling step: the reference then reads the
                          # neighbouring profile, _extcoeff.c:302-306 -- a documented deviation)
                          wnosamp=osamp, nlor=int(rng.integers(10, 16)), ndop=int(rng.integers(4, 7)),
                          extent=float(rng.choice([8.0, 40.0])), cutoff=float(rng.choice([0.0, 3.0])),
                          niso=niso, seed=seed)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    size_a = np.array(vg['size'], np.int64)                    # voigt.py:109-110 uses int
    size_b = np.array(vg['size'], np.int64)
    idx_a, idx_b = np.zeros_like(size_a), np.zeros_like(size_b)
    n = int(np.sum(2 * np.maximum(size_a, 1) + 1)) + 16
    prof_a, prof_b = np.zeros(n), np.zeros(n)
    assert V.grid(prof_a, size_a, idx_a, vg['lorentz'], vg['doppler'], g['ownstep'], 0) == 1
    Vr.grid(prof_b, size_b, idx_b, vg['lorentz'], vg['doppler'], g['ownstep'], 0)
    assert np.array_equal(size_a, size_b) and np.array_equal(idx_a, idx_b), 'vprofile sizes'
    np.testing.assert_allclose(prof_a, prof_b, rtol=2e-12, err_msg='vprofile.grid')
    add = int(rng.random() < 0.5)
    isoiext = np.array(iso['isoiext'], np.int64)
    if not add:
        isoiext = rng.integers(0, 2, niso).astype(np.int64)
        isoiext[0] = 0
    rows = 1 if add else int(isoiext.max()) + 1
    ethresh = float(rng.choice([1e-30, 1e-4]))
    nw = g['nwave']
    for k in range(atm['nlayers']):
        a, b = np.zeros((rows, nw)), np.zeros((rows, nw))
        args = (prof_b, size_b, idx_b, vg['lorentz'], vg['doppler'], g['wn'], g['own'],
                np.array(g['divisors'], np.int64), atm['dens'][k], atm['mol_radius'],
                atm['mol_mass'], np.array(iso['isoimol'], np.int64), iso['isomass'],
                iso['isoratio'], iso['isoz'][:, k].copy(), isoiext, ln['lwn'], ln['elow'],
                ln['gf'], np.array(ln['lid'], np.int64), vg['cutoff'], ethresh,
                float(atm['temp'][k]), 0, add, 0)
        assert E.extinction(a, *args) == 1
        Er.extinction(b, *args)
        tag = (f'extinction layer {k} of {atm["nlayers"]}: add={add} rows={rows} ethresh={ethresh} '
               f'cutoff={vg["cutoff"]} niso={niso} isoiext={isoiext.tolist()} nw={nw} '
               f'nlines={len(ln["lwn"])} osamp={osamp}')
        if not np.array_equal(a == 0, b == 0) or not np.allclose(a, b, rtol=1e-10, atol=0):
            bad = np.nonzero(~np.isclose(a, b, rtol=1e-10, atol=0))
            # a fresh plan for this layer alone: is it the reuse of the cached plan?
            E.invalidate()
            c = np.zeros((rows, nw))
            E.extinction(c, *args)
            from pyratbay_amd import engine as eng
            res = {}
            for name, vt in (('own table', eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], osamp)),
                             ('from_flat', eng.VoigtTable.from_flat(prof_b, size_b, idx_b, vg['lorentz'], vg['doppler'], osamp))):
                ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
                for ml in (atm['nlayers'], 1):
                    lb = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                                 iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext.astype(np.int32),
                                 vg['cutoff'], ethresh, max_layers=ml)
                    e = lb.extinction(eng.dev(atm['temp'][k:k + 1]), eng.dev(atm['dens'][k:k + 1]),
                                      eng.dev(iso['isoz'][:, k:k + 1].copy()), add=bool(add)).cpu().numpy()[0]
                    res[f'{name} max_layers={ml}'] = bool(np.allclose(e, b, rtol=1e-10, atol=0))
            raise AssertionError(f'{tag}: {len(bad[0])} samples differ (rows {np.unique(bad[0]).tolist()}); engine: {res}; '
                                 f'fresh plan equal to reference: {np.allclose(c, b, rtol=1e-10, atol=0)}; '
                                 f'first diff a={a[bad][0]:.6e} ref={b[bad][0]:.6e}')
    E.invalidate()


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import ref
    if not ref.available():
        sys.exit('oracle/_ref is not built')
    from pyratbay_amd import engine
    engine.require_gpu()
    import importlib
    lib = {n: importlib.import_module('pyratbay_amd.lib.' + n)
           for n in ('_trapezoid', '_blackbody', '_simpson', 'cutils', '_indices', '_extcoeff',
                     'vprofile')}
    bad = []
    for seed in range(count):
        try:
            one(np.random.default_rng(4000 + seed), lib, ref)
            if seed % 5 == 0:
                heavy(np.random.default_rng(90000 + seed), lib, ref, seed)
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
