"""Randomised sweep of the round-5 code paths (GPU box):
  (1) the layers nobody reads: random batches (1-128 impact parameters, 1-6 species, ragged
      widths, itop, walkers that differ from the base model by per cent or by orders of
      magnitude), RANDOM tile limits per block of 256 ordered columns (from exact to far too
      low): limited interpolation + limited transit / emission + the device-gated repair must
      give the spectra of the unlimited ordered path BIT FOR BIT, the flags must name exactly
      the walkers with a wavefront (transit) / a column (emission) open beyond its limit, and the
      interpolation must leave the layers above itop untouched;
  (2) TableSpectrum.eval_bands with random tile margins (0 ... 6) against column_order=None;
  (3) stacked atmospheres (dist.StackedShard / ShardPipeline(stack=K)): K random atmospheres
      per call on a random shard against LBLSpectrum.run() of each (1e-12), one rank;
  (4) line lists whose isotopes interleave (accepted, equal to the oracle's sequential pass) and
      lists that step back within an isotope (refused).
  (5) what follows the radiative transfer on the emission-type paths (engine.emission_observables,
      PassBands.set_eclipse / star_bandflux / per-walker dilution): random fluxes of 1 ... 5000
      samples with zeros, infinities and NaNs planted in flux and stellar flux, against the
      oracle's NumPy restatement BIT FOR BIT (band fluxes 1e-13).
usage: python tools/fuzz_r5.py [count] [seed0]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def host(t):
    return t.cpu().numpy()


def limited_case(eng, rng):
    import torch
    L = int(rng.choice([2, 3, 9, 16, 17, 33, 48, 64, 80, 81, 100, 128]))
    itop = int(rng.integers(0, max(1, L // 4)))
    if L - itop < 2:
        itop = 0
    W = int(rng.choice([2, 31, 64, 255, 256, 257, 700, 1601, 3000]))
    nspec, ntemp, nw = int(rng.integers(1, 7)), int(rng.integers(2, 8)), int(rng.integers(1, 13))
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    spread = float(rng.choice([0.0, 0.05, 1.5]))           # walkers near / far from the base
    dens *= 10.0**(spread * rng.uniform(-1, 1, (nw, 1, 1)))
    maxdepth = float(rng.choice([10.0, 0.3, np.inf]))
    et, tt, td, dd = eng.dev(etable), eng.dev(ttable), eng.dev(temps), eng.dev(dens)
    rad = eng.dev(np.tile(radius0, (nw, 1)) * (1 + 0.01 * rng.uniform(-1, 1, (nw, 1))))
    emission = rng.random() < 0.4
    ec0 = eng.interp_ec_batch(et, tt, td, dd)
    if emission:
        wn = eng.dev(np.linspace(4000.0, 4000.0 + 0.05 * (W - 1), W))
        mu, wts = eng.default_quadrature()
        mu, wts = eng.dev(mu), eng.dev(wts)
        intervals = (rad[:, :-1] - rad[:, 1:]).contiguous()
        ideep = torch.stack([eng.plane_parallel_optical_depth(
            ec0[w], intervals[w].contiguous(), itop, L, maxdepth)[1] for w in range(nw)])
    else:
        path = eng.transit_path_device(rad, itop)
        _, _, ideep = eng.transit_spectrum_batch(ec0, path, rad, 8.8e10, itop, L, maxdepth,
                                                 want_depth=True)
    order = torch.sort(ideep[0], stable=True).indices
    eto = et[..., order].contiguous()
    col = order.to(torch.int32)
    full = eng.interp_ec_batch(eto, tt, td, dd)
    if emission:
        wno = wn[order].contiguous()
        want = eng.emission_flux_batch(full, intervals, wno, td, mu, wts, itop, L, maxdepth, col)
    else:
        want = eng.transit_spectrum_ordered(full, path, rad, col, 8.8e10, itop, L, maxdepth)
    nblk = -(-W // 256)
    ntiles = -(-(L - itop) // 16)
    mode = rng.integers(0, 3)
    if mode == 0:
        tile = rng.integers(0, ntiles, nblk)
    elif mode == 1:
        tile = np.zeros(nblk, int)
    else:
        deepest = host(ideep[:, order].max(dim=0).values)
        pad = np.concatenate([deepest, np.full(nblk * 256 - W, deepest[-1])])
        tile = np.clip((pad.reshape(nblk, 256).max(1) - itop + int(rng.integers(0, 4))) // 16, 0,
                       ntiles - 1)
    tile_d = eng.dev(tile, torch.int32)
    flags = torch.zeros(nw + 1, dtype=torch.int32, device='cuda')
    ec = torch.full((nw, L, W), -7.0, dtype=torch.float64, device='cuda')
    iwork = torch.empty(nw * L * 17 + 8, dtype=torch.float64, device='cuda')
    eng.interp_ec_batch(eto, tt, td, dd, out=ec, tile_limit=tile_d, row0=itop, work=iwork)
    written = ec != -7.0
    lay = torch.arange(L, device='cuda')[None, :, None]
    blk = (torch.arange(W, device='cuda') // 256)[None, None, :]
    must = (lay >= itop) & (lay <= itop + 16 * (tile_d.to(torch.int64)[blk] + 1) - 1)
    assert bool((written | ~must).all()), 'a wanted layer was not written'
    assert torch.equal(ec[written], full[written]), 'a written layer differs'
    assert not bool(written[:, :itop].any()), 'a layer above itop was written'
    klim = itop + 16 * (tile + 1) - 1
    idp = host(ideep[:, order])
    if emission:
        got = eng.emission_flux_batch(ec, intervals, wno, td, mu, wts, itop, L, maxdepth, col,
                                      tile_limit=tile_d, flags=flags)
        # a column is open beyond its limit when its loop would read layer klim + 1
        over = np.array([(np.minimum(idp[w], L - 1) > klim[np.arange(W) // 256]).any()
                         for w in range(nw)])
    else:
        twork = torch.empty(eng._capi.lib().pb_transit_work_doubles(L, itop, L, W, nw),
                            dtype=torch.float64, device='cuda')
        got = eng.transit_spectrum_ordered(ec, path, rad, col, 8.8e10, itop, L, maxdepth,
                                           tile_limit=tile_d, flags=flags, work=twork)
        starts = np.arange(0, W, 32)
        wtile = (idp - itop) // 16
        over = np.array([(np.maximum.reduceat(wtile[w], starts) > tile[starts // 256]).any()
                         for w in range(nw)])
    f = host(flags)
    assert f[nw] == int(f[:nw].any()), 'the any-flag disagrees'
    assert np.array_equal(f[:nw] != 0, over), ('flags', emission, f, over)
    eng.interp_ec_batch(eto, tt, td, dd, out=ec, gate=flags[nw:nw + 1], work=iwork)
    if emission:
        eng.emission_flux_batch(ec, intervals, wno, td, mu, wts, itop, L, maxdepth, col,
                                gate=flags, out=got)
    else:
        eng.transit_spectrum_ordered(ec, path, rad, col, 8.8e10, itop, L, maxdepth, gate=flags,
                                     out=got, work=twork)
    assert torch.equal(got, want), 'limited + repair differs from the unlimited path'
    return int(f[nw])


def eval_bands_case(eng, rng):
    import torch
    from pyratbay_amd import synth
    L = int(rng.choice([9, 33, 80, 100]))
    W = int(rng.choice([64, 300, 1025, 2049]))
    nspec, ntemp, nw = int(rng.integers(1, 5)), 6, int(rng.integers(2, 20))
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    dens *= 10.0**(float(rng.choice([0.02, 1.0])) * rng.uniform(-1, 1, (nw, 1, 1)))
    pb = eng.PassBands(g['wn'], [(1, np.ones(W - 2), 1.0)])
    rt = 'emission' if rng.random() < 0.3 else 'transit'
    res = {}
    for order in (None, 'auto'):
        m = eng.TableSpectrum(etable, ttable, g['wn'], radius0, 8.8e10, rt_path=rt,
                              column_order=order)
        m.tile_margin = int(rng.integers(0, 7))
        res[order] = m.eval_bands(eng.dev(temps), eng.dev(dens), pb, chunk=int(rng.choice([5, 64])))
    assert torch.equal(res[None], res['auto']), f'{rt}: band fluxes depend on the tile limits'


def stacked_case(eng, rng, seed):
    import torch
    from pyratbay_amd import synth
    from pyratbay_amd.dist import ShardPipeline
    K = int(rng.integers(2, 5))
    nwave, nl = int(rng.integers(1500, 6000)), int(rng.integers(3, 12))
    case = synth.lbl_case(nwave, nl, int(rng.integers(2000, 20000)), wnosamp=24, nlor=14, ndop=7,
                          extent=60.0, cutoff=float(rng.choice([2.0, 4.0])),
                          niso=int(rng.integers(1, 4)), seed=seed)
    atm, iso = case['atm'], case['iso']
    nwave = case['grid']['nwave']
    a = int(rng.integers(0, nwave // 2))
    b = int(rng.integers(a + 10, nwave + 1))
    serial = eng.LBLSpectrum(case, rt_path='transit', wbegin=a, wcount=b - a, timestamps=False)
    atms, want = [], []
    for k in range(K):
        temp = atm['temp'] * (1.0 + 0.05 * rng.uniform(-1, 1)) + rng.uniform(0, 5)
        dens = atm['dens'] * (atm['temp'] / temp)[:, None] * 10.0**rng.uniform(-0.3, 0.3)
        isoz = iso['isoz'] * (1.0 + 0.02 * rng.uniform(-1, 1))
        radius = atm['radius'] * (1.0 + 0.003 * rng.uniform(-1, 1))
        atms.append((temp, dens, isoz, radius))
        serial.set_atmosphere(*atms[-1])
        want.append(serial.run().clone())
    pipe = ShardPipeline(case, 1, 0, depth=int(rng.integers(1, 4)), voigt=serial.voigt,
                         lines=serial.lines, stack=K)
    for m in pipe.models:
        if rng.random() < 0.5:
            m.kmax_exchange = lambda t: None
        for k, t in enumerate(atms):
            m.set_atmosphere(k, *t)
    outs = []
    for i in range(int(rng.integers(1, 4))):
        r = pipe.submit()
        if r is not None:
            r[1].synchronize()
            outs.append([x.clone() for x in r[0]])
    last = pipe.flush()
    torch.cuda.synchronize()
    outs.append([x.clone() for x in last[0]])
    for o in outs:
        for k in range(K):
            np.testing.assert_allclose(host(o[k][a:b]), host(want[k]), rtol=1e-12)


def order_case(eng, orc, rng, seed):
    from pyratbay_amd import _capi, synth
    case = synth.lbl_case(int(rng.integers(1500, 4000)), 3, int(rng.integers(500, 4000)),
                          wnosamp=24, nlor=12, ndop=6, extent=60.0, cutoff=3.0,
                          niso=int(rng.integers(2, 4)), seed=seed)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    o = np.argsort(ln['lwn'], kind='stable')               # isotopes interleaved, each ascending
    mixed = {k: np.ascontiguousarray(ln[k][o]) for k in ('lwn', 'elow', 'gf', 'lid')}
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], g['wnosamp'])
    ll = eng.LineList(mixed['lwn'], mixed['elow'], mixed['gf'], mixed['lid'], len(iso['isomass']),
                      g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'],
                  case['ethresh'], max_layers=3)
    ext = host(lbl.extinction(eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])))
    profile = vt.flat()
    layer = int(rng.integers(0, 3))
    want = np.zeros((1, g['nwave']))
    orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                   g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                   atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                   iso['isoz'][:, layer].copy(), iso['isoiext'], mixed['lwn'], mixed['elow'],
                   mixed['gf'], mixed['lid'], vg['cutoff'], case['ethresh'], atm['temp'][layer],
                   0, 1, 0)
    assert np.array_equal(ext[layer] == 0, want == 0)
    np.testing.assert_allclose(ext[layer], want, rtol=1e-10)
    # two in-range lines of one isotope swapped: refused
    inr = np.flatnonzero((mixed['lwn'] > g['own'][0]) & (mixed['lwn'] < g['own'][-1]) &
                         (mixed['lid'] == mixed['lid'][len(o) // 2]))
    if len(inr) >= 2 and mixed['lwn'][inr[0]] != mixed['lwn'][inr[-1]]:
        bad = mixed['lwn'].copy()
        bad[[inr[0], inr[-1]]] = bad[[inr[-1], inr[0]]]
        try:
            eng.LineList(bad, mixed['elow'], mixed['gf'], mixed['lid'], len(iso['isomass']), g['own'])
        except _capi.PbError as e:
            assert 'ascending wavenumber order' in str(e)
        else:
            raise AssertionError('a list that steps back within an isotope was accepted')


def same_bits(a, b):
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b)))) and \
        bool(np.all(np.signbit(a) == np.signbit(b)))


def observables_case(eng, orc, rng):
    W = int(rng.choice([1, 2, 63, 64, 65, 257, 1000, 5000]))
    wn = np.sort(rng.uniform(500.0, 30000.0, W))
    flux = 10.0**rng.uniform(-3, 7, W)
    star = 10.0**rng.uniform(4, 8, W)
    for arr in (flux, star):
        for val in (0.0, -0.0, np.inf, np.nan, 5e-324, 1e308):
            if rng.random() < 0.3:
                arr[rng.integers(0, W)] = val
    rplanet, rstar, distance = 10.0**rng.uniform(8, 10), 10.0**rng.uniform(10, 11), 10.0**rng.uniform(18, 21)
    dil = None if rng.random() < 0.4 else float(rng.uniform(0.0, 1.0))
    dflux, dstar, dwn = eng.dev(flux), eng.dev(star), eng.dev(wn)
    with np.errstate(all='ignore'):
        for kind in ('emission', 'eclipse'):
            got, fp = eng.emission_observables(dflux, kind, dstar, rplanet, rstar, dil)
            want, wfp = orc.emission_observables(flux, kind, star, rplanet, rstar, dil)
            assert same_bits(host(got), want) and same_bits(host(fp), wfp), kind
        got, fp = eng.emission_observables(dflux, 'f_lambda', rplanet=rplanet, f_dilution=dil,
                                           wn=dwn, distance=distance)
        _, wfp = orc.emission_observables(flux, 'emission', f_dilution=dil)
        assert same_bits(host(got), orc.f_lambda_units(wfp, wn, rplanet, distance))
        assert same_bits(host(fp), wfp)
    assert same_bits(host(dflux), flux)                       # never in place unless asked
    # bands: eclipse factor and per-walker dilution on finite spectra
    if W >= 8:
        nb = int(rng.integers(1, 6))
        bands = []
        for _ in range(nb):
            lo = int(rng.integers(0, W - 2))
            hi = int(rng.integers(lo + 2, W + 1))
            bands.append((lo, rng.uniform(0.0, 1.0, hi - lo), float(rng.uniform(0.1, 3.0))))
        pb = eng.PassBands(wn, bands)
        star_f = 10.0**rng.uniform(4, 8, W)
        sb = pb.star_bandflux(star_f)
        want_sb = np.array([np.trapezoid(star_f[s:s + len(r)] * r, wn[s:s + len(r)]) * h
                            for s, r, h in bands])
        np.testing.assert_allclose(sb, want_sb, rtol=1e-13)
        nw = int(rng.integers(1, 5))
        spectra = 10.0**rng.uniform(-3, 7, (nw, W))
        fd = rng.uniform(0.1, 1.0, nw)
        pb.set_eclipse(rplanet, rstar, sb)
        got = host(pb.integrate_batch(eng.dev(spectra), f_dilution=eng.dev(fd)))
        plain = np.array([[np.trapezoid(sp[s:s + len(r)] * r, wn[s:s + len(r)]) * h
                           for s, r, h in bands] for sp in spectra])
        want = orc.eclipse_bandflux(plain * fd[:, None], rplanet, rstar, sb)
        np.testing.assert_allclose(got, want, rtol=1e-13)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 51000
    from pyratbay_amd import engine as eng
    from oracle import oracle as orc
    eng.require_gpu()
    fails, repaired, n = 0, 0, [0, 0, 0, 0, 0]
    for i in range(count):
        seed = seed0 + i
        rng = np.random.default_rng(seed)
        try:
            kind = i % 8
            if i % 16 == 3:
                observables_case(eng, orc, rng)
                n[4] += 1
            elif kind < 4:
                repaired += limited_case(eng, rng)
                n[0] += 1
            elif kind < 6:
                eval_bands_case(eng, rng)
                n[1] += 1
            elif kind == 6:
                stacked_case(eng, rng, seed)
                n[2] += 1
            else:
                order_case(eng, orc, rng, seed)
                n[3] += 1
        except Exception:                                       # noqa: BLE001
            fails += 1
            print(f'FAIL seed {seed}')
            traceback.print_exc()
    print(f'fuzz_r5: {count} cases from seed {seed0}: {fails} failures; tile-limited batches '
          f'{n[0]} ({repaired} with a repair pass), eval_bands with random margins {n[1]}, stacked '
          f'shards {n[2]}, interleaved / stepping-back line lists {n[3]}, observables {n[4]}')
    print('failures: []' if not fails else f'failures: {fails}')
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
