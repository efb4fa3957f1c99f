"""Randomised sweep of the round-3 code paths (GPU box):
  (1) chunked line lists (a random record budget) against the one-call form, bit for bit, both with
      PB_STAGE_SPLIT=1 -- add 0/1, several rows, skipped isotopes, ethresh up to 1e-2, long rows,
      wavenumber shards;
  (2) per-layer phase split (random PB_STAGE_DEEP) against the unsplit launch at 1e-13, equal zero
      pattern, and against the compiled reference at 1e-10;
  (3) two-phase shard calls with the window map in LDS, in global memory (PB_WM_LDS_CAP=0) and
      switched off: identical maxima and sums;
  (4) the matrix-core transit kernel against the vector kernel (1e-12) for random shapes;
  (5) LBLSpectrum timestamps: keys and positivity.
usage: python tools/fuzz_r3.py [count] [seed0]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def host(t):
    return t.cpu().numpy()


def ext_case(eng, ref, rng, seed):
    import torch
    from pyratbay_amd import synth
    long_rows = rng.random() < 0.25
    nwave = int(rng.integers(300, 6000))
    nlayers = int(rng.integers(1, 12))
    nlines = int(rng.integers(200, 12000))
    niso = int(rng.integers(1, 5))
    if long_rows:
        kw = dict(wnosamp=12, nlor=8, ndop=4, extent=float(rng.choice([1500.0, 4000.0])),
                  cutoff=float(rng.choice([40.0, 80.0])))
    else:
        kw = dict(wnosamp=int(rng.choice([6, 12, 24, 60])), nlor=10, ndop=5,
                  extent=float(rng.choice([8.0, 40.0, 150.0])),
                  cutoff=float(rng.choice([0.5, 3.0, 30.0])))
    case = synth.lbl_case(nwave, nlayers, nlines, wnstep=float(rng.choice([0.01, 0.05, 0.2])),
                          niso=niso, seed=seed, ptop=10.0**rng.uniform(-7, -4),
                          pbottom=10.0**rng.uniform(-1, 2), **kw)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    add = bool(rng.random() < 0.5)
    ethresh = float(rng.choice([1e-30, 1e-6, 1e-2]))
    isoiext = np.array(iso['isoiext'], np.int32).copy()
    if not add:
        isoiext = rng.integers(0, int(rng.integers(1, 4)), niso).astype(np.int32)
        isoiext[rng.integers(0, niso)] = 0
    if niso > 1 and rng.random() < 0.3:
        isoiext[rng.integers(1, niso)] = -1
    rows = 1 if add else int(isoiext.max()) + 1
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'], True)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])

    def plan():
        p = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                    iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext, vg['cutoff'],
                    ethresh, max_layers=nlayers)
        p.set_gather_mode('staged')
        return p
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    info = dict(chunks=0, deep=False, wm=False)
    # (1) chunks
    os.environ['PB_STAGE_SPLIT'] = '1'
    os.environ.pop('PB_STAGE_DEEP', None)
    whole = plan()
    want = whole.extinction(t, d, z, add=add).clone()
    if whole.last_gather_kernel.endswith('k_ext_staged') and ll.ngroups > 0:
        cut = plan()
        total = ll.ngroups * nlayers * 16
        for parts in (rng.uniform(1.5, 4.0), rng.uniform(4.0, 12.0)):
            budget = int(total / parts)
            while True:
                cut.set_record_budget(max(budget, 16))
                try:
                    got = cut.extinction(t, d, z, add=add)
                    break
                except Exception as e:                     # a key larger than the budget
                    if 'record budget' not in str(e):
                        raise
                    budget = int(budget * 1.5) + 64
            assert torch.equal(got, want), f'chunked ({cut.last_chunks} chunks) != one call'
            info['chunks'] = max(info['chunks'], cut.last_chunks)
        if nwave > 600:
            a = int(rng.integers(0, nwave - 300))
            b = int(rng.integers(a + 1, nwave))
            part = cut.extinction(t, d, z, add=add, wbegin=a, wcount=b - a)
            assert torch.equal(part, want[:, :, a:b]), 'chunked shard != slice of the one call'
        cut.close()
        # (3) window map variants of a two-phase shard
        if nwave > 600:
            a = int(rng.integers(0, nwave // 2))
            b = int(rng.integers(a + 1, nwave))
            res = []
            for env in ({}, {'PB_WM_LDS_CAP': '0'}, {'PB_NO_WINDOW_MAP': '1'}):
                for k in ('PB_WM_LDS_CAP', 'PB_NO_WINDOW_MAP'):
                    os.environ.pop(k, None)
                os.environ.update(env)
                p2 = plan()
                e2 = p2.extinction_begin(t, d, z, add=add, wbegin=a, wcount=b - a)
                km = p2.kmax_tensor().clone()
                p2.kmax_tensor().copy_(whole.kmax_tensor())        # the all-reduced maxima
                p2.extinction_end()
                res.append((km, e2.clone()))
                p2.close()
            for k in ('PB_WM_LDS_CAP', 'PB_NO_WINDOW_MAP'):
                os.environ.pop(k, None)
            for km, e2 in res[1:]:
                assert torch.equal(km, res[0][0]) and torch.equal(e2, res[0][1]), 'window map forms differ'
            assert torch.equal(res[0][1], want[:, :, a:b]), 'two-phase shard != slice of the one call'
            info['wm'] = True
        # (2) per-layer split
        os.environ.pop('PB_STAGE_SPLIT', None)
        os.environ['PB_STAGE_DEEP'] = '0'
        base = whole.extinction(t, d, z, add=add).clone()
        os.environ['PB_STAGE_DEEP'] = f'{rng.uniform(0.1, 1.0):.2f},{int(rng.integers(2, 9))}'
        got = whole.extinction(t, d, z, add=add)
        assert torch.equal(got == 0, base == 0), 'deep split: zero pattern'
        rel = ((got - base).abs() / base.abs().clamp_min(1e-300)).max().item()
        assert rel <= 1e-13, f'deep split differs by {rel:.2e}'
        os.environ.pop('PB_STAGE_DEEP', None)
        info['deep'] = True
        gh = host(got)
        profile, psize, pindex = vt.flat(), np.array(vt.size), np.array(vt.index)
        E = ref.module('_extcoeff')
        for k in range(nlayers):
            w = np.zeros((rows, g['nwave']))
            E.extinction(w, profile, psize, pindex, vg['lorentz'], vg['doppler'], g['wn'],
                         g['own'], g['divisors'], atm['dens'][k], atm['mol_radius'],
                         atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                         iso['isoz'][:, k].copy(), isoiext, ln['lwn'], ln['elow'], ln['gf'],
                         ln['lid'], vg['cutoff'], ethresh, float(atm['temp'][k]), 0, int(add), 0)
            assert np.array_equal(gh[k] == 0, w == 0), 'vs reference: zero pattern'
            np.testing.assert_allclose(gh[k], w, rtol=1e-10)
    os.environ.pop('PB_STAGE_SPLIT', None)
    whole.close()
    ll.close()
    vt.close()
    return info


def transit_case(eng, rng, seed):
    import cases
    L = int(rng.integers(2, 129))
    W = int(rng.integers(2, 700))
    nw = int(rng.integers(2, 6))
    itop = int(rng.integers(0, max(1, L // 3)))
    ibottom = L if rng.random() < 0.6 else int(rng.integers(itop + 1, L + 1))
    maxdepth = float(rng.choice([10.0, 1.0, np.inf]))
    c = cases.column_case(seed=seed, nlayers=L, nwave=W)
    scale = 10.0**rng.uniform(-3, 3, W)
    ecs = np.array([c['ec'] * scale * 10.0**rng.uniform(-0.3, 0.3) for _ in range(nw)])
    radius = np.array([np.sort(c['radius'] * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    rad_d = eng.dev(radius)
    path = eng.transit_path_device(rad_d, itop)
    ec_d = eng.dev(ecs)
    os.environ['PB_TRANSIT_MFMA'] = '1'
    got = host(eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, maxdepth))
    os.environ['PB_TRANSIT_MFMA'] = '0'
    vec = host(eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, maxdepth))
    os.environ.pop('PB_TRANSIT_MFMA')
    assert np.all(np.isfinite(got)), 'matrix-core transit: non-finite spectrum'
    np.testing.assert_allclose(got, vec, rtol=1e-12)


def table_transit_case(eng, rng, seed):
    """pb_table_transit_batch (interpolation + optical depth + transmission in one pass) against
    pb_interp_ec_batch + pb_transit_spectrum_batch on random shapes: 1-8 species, 2-128 layers,
    ragged grids, itop, temperatures on nodes and at the table's ends (1e-12)."""
    nmol = int(rng.integers(1, 9))
    ntemp = int(rng.integers(2, 9))
    L = int(rng.integers(2, 129))
    W = int(rng.integers(2, 600))
    nw = int(rng.integers(1, 7))
    itop = int(rng.integers(0, max(1, L // 3)))
    if L - itop < 2:
        itop = 0
    if not eng.table_transit_supported(nmol, ntemp, L, itop, L, W):
        return
    ttable = np.sort(rng.uniform(300.0, 3000.0, ntemp))
    ttable[0], ttable[-1] = 300.0, 3000.0
    press = np.logspace(-6, 2, L)
    colscale = 10.0**rng.uniform(-33.5, -27.0, W)
    etable = 10.0**rng.uniform(-0.3, 0.3, (nmol, ntemp, L, W)) * colscale
    temps = rng.uniform(300.0, 3000.0, (nw, L))
    temps[0] = ttable[rng.integers(0, ntemp, L)]
    if nw > 1:
        temps[1] = rng.choice([300.0, 3000.0])
    dens = press[None, :, None]**0.9 * 10.0**rng.uniform(17, 18, (nw, L, nmol))
    radius = np.array([np.sort(np.linspace(8.0e9, 7.0e9, L) * (1 + 0.01 * rng.uniform(-1, 1)) +
                               rng.uniform(-1e5, 1e5, L))[::-1] for _ in range(nw)])
    maxdepth = float(rng.choice([10.0, 1.0, np.inf]))
    et, tt = eng.dev(etable), eng.dev(ttable)
    td, dd, rd = eng.dev(temps), eng.dev(dens), eng.dev(radius)
    path = eng.transit_path_device(rd, itop)
    os.environ['PB_TT_PAIR'] = str(rng.choice(['0', '1', '2', '2']))
    one = host(eng.table_transit_batch(et, tt, td, dd, path, rd, 8.8e10, itop, L, maxdepth))
    os.environ.pop('PB_TT_PAIR')
    ec = eng.interp_ec_batch(et, tt, td, dd)
    two = host(eng.transit_spectrum_batch(ec, path, rd, 8.8e10, itop, L, maxdepth))
    assert np.all(np.isfinite(one)), 'one-pass table transit: non-finite spectrum'
    np.testing.assert_allclose(one, two, rtol=1e-12)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
    from oracle import ref
    if not ref.available():
        sys.exit('oracle/_ref is not built')
    from pyratbay_amd import engine
    engine.require_gpu()
    bad, chunks, deep, wm = [], 0, 0, 0
    for i in range(count):
        seed = seed0 + i
        try:
            info = ext_case(engine, ref, np.random.default_rng(seed), seed)
            chunks += info['chunks'] > 1
            deep += info['deep']
            wm += info['wm']
            for j in range(3):
                transit_case(engine, np.random.default_rng(seed * 7 + j), seed * 7 + j)
                table_transit_case(engine, np.random.default_rng(seed * 11 + j), seed * 11 + j)
        except Exception:                                  # noqa: BLE001
            bad.append(seed)
            print('FAIL seed', seed)
            traceback.print_exc(limit=4)
        if i % 50 == 49:
            print(f'{i + 1} seeds, {len(bad)} failures', flush=True)
    print(f'{count} seeds: {chunks} chunked, {deep} with a per-layer split, {wm} window-map shards, '
          f'{3 * count} transit batches, {3 * count} one-pass table batches')
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
