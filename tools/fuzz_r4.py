"""Randomised sweep of the round-4 code paths (GPU box), every case against the compiled reference
(oracle/_ref, rtol 1e-10, identical zero pattern):
  (1) the wave-autonomous kernel (gather mode 'wave') on random grids -- rows below and above its
      384-sample limit in one atmosphere, add 0/1, several rows, skipped isotopes, ethresh up to
      1e-2 -- against the staged kernel alone (1e-12) and bit-exact wavenumber shards;
  (2) band-structured line lists (synth.band_positions, random contrast / band count / duplicate
      share) in automatic mode: the per-tile phase split and the sparse tiles' global gather;
      two calls bitwise equal; PB_TILE_SPLIT=0 / PB_TILE_GLOBAL=0 agree to 1e-12;
  (3) partition functions: engine.PartitionTable against tli.iso_partition on random tables,
      bit for bit.
usage: python tools/fuzz_r4.py [count] [seed0]"""
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
    banded = rng.random() < 0.5
    nwave = int(rng.integers(2500, 14000))
    nlayers = int(rng.integers(2, 10))
    nlines = int(rng.integers(2000, 60000))
    if banded and rng.random() < 0.5:
        # large enough for the automatic mode to choose the staged kernel (>= 750 workgroups):
        # the per-tile split and the sparse tiles' global gather
        nwave = int(rng.integers(40000, 90000))
        nlayers = int(rng.integers(8, 13))
        nlines = int(rng.integers(100000, 300000))
    niso = int(rng.integers(1, 4))
    osamp = int(rng.choice([12, 24, 36, 60]))
    wnstep = float(rng.choice([0.02, 0.05, 0.1]))
    kw = dict(wnosamp=osamp, nlor=12, ndop=6, extent=float(rng.choice([60.0, 150.0, 300.0])),
              cutoff=float(rng.choice([0.1, 0.25, 0.5]) * 384 * wnstep))
    if banded and rng.random() < 0.35:
        # a launch whose base phase split is below 8 (>= 250 (tile, layer) pairs) with rows too
        # long for the resident kernel: the per-tile split and the sparse tiles' global gather
        nwave = int(rng.integers(100000, 170000))
        nlayers = int(rng.integers(10, 15))
        nlines = int(rng.integers(150000, 400000))
        osamp = int(rng.choice([24, 36]))
        kw = dict(wnosamp=osamp, nlor=12, ndop=6, extent=300.0,
                  cutoff=float(rng.uniform(0.5, 1.2) * 384 * wnstep))
    bands = None
    if banded:
        bands = dict(nbands=int(rng.integers(1, 6)), contrast=float(10.0**rng.uniform(1.5, 3.0)),
                     in_bands=float(rng.uniform(0.6, 0.95)), duplicates=float(rng.uniform(0, 0.1)))
    case = synth.lbl_case(nwave, nlayers, nlines, wnstep=wnstep, niso=niso, seed=seed,
                          ptop=10.0**rng.uniform(-7, -4), pbottom=10.0**rng.uniform(-1, 2),
                          bands=bands, **kw)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    add = bool(rng.random() < 0.6)
    ethresh = float(rng.choice([1e-30, 1e-6, 1e-2]))
    isoiext = np.array(iso['isoiext'], np.int32).copy()
    if not add:
        isoiext = rng.integers(0, int(rng.integers(1, 3)), niso).astype(np.int32)
        isoiext[rng.integers(0, niso)] = 0
    if niso > 1 and rng.random() < 0.3:
        isoiext[rng.integers(1, niso)] = -1
    rows = 1 if add else int(isoiext.max()) + 1
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], osamp, True)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext, vg['cutoff'],
                  ethresh, max_layers=nlayers)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    info = dict(wave=0, tiles=False, banded=banded)
    profile, psize, pindex = vt.flat(), np.array(vt.size), np.array(vt.index)
    E = ref.module('_extcoeff')
    want = np.zeros((nlayers, rows, g['nwave']))
    for k in range(nlayers):
        E.extinction(want[k], profile, psize, pindex, vg['lorentz'], vg['doppler'], g['wn'],
                     g['own'], g['divisors'], atm['dens'][k], atm['mol_radius'],
                     atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                     iso['isoz'][:, k].copy(), isoiext, ln['lwn'], ln['elow'], ln['gf'],
                     ln['lid'], vg['cutoff'], ethresh, float(atm['temp'][k]), 0, int(add), 0)

    def check(name, got):
        gh = host(got)
        bad = ~np.isclose(gh, want, rtol=1e-10, atol=0.0)
        if bad.any():
            idx = np.argwhere(bad)
            raise AssertionError(
                f'{name} vs reference: {bad.sum()} samples, layers {np.unique(idx[:, 0])[:8]}, '
                f'samples {idx[:, 2].min()}..{idx[:, 2].max()}, first {gh[tuple(idx[0])]!r} '
                f'want {want[tuple(idx[0])]!r}; kernel {lbl.last_gather_kernel}')

    lbl.set_gather_mode('staged')
    staged = lbl.extinction(t, d, z, add=add).clone()
    ok_staged = lbl.last_gather_kernel == 'k_ext_staged'
    check('staged', staged)
    from pyratbay_amd import _capi
    if ok_staged and _capi.experiments():                 # (the wave kernel: libpbhip_exp.so only)
        lbl.set_gather_mode('wave')
        wv = lbl.extinction(t, d, z, add=add).clone()
        info['wave'] = int(lbl.last_wave_layers(nlayers).sum())
        check('wave', wv)
        assert torch.equal(lbl.extinction(t, d, z, add=add), wv), 'wave: two runs differ'
        rel = ((wv - staged).abs() / staged.abs().clamp_min(1e-300)).max().item()
        assert torch.equal(wv == 0, staged == 0) and rel <= 1e-12, f'wave vs staged {rel:.2e}'
        a = int(rng.integers(0, nwave - 600))
        b = int(rng.integers(a + 1, nwave))
        part = lbl.extinction(t, d, z, add=add, wbegin=a, wcount=b - a)
        # (a shard's launch may choose another phase split: pin it for the exactness check)
        os.environ['PB_STAGE_SPLIT'] = '2'
        full2 = lbl.extinction(t, d, z, add=add).clone()
        part = lbl.extinction(t, d, z, add=add, wbegin=a, wcount=b - a)
        os.environ.pop('PB_STAGE_SPLIT')
        assert torch.equal(part, full2[:, :, a:b]), 'wave: shard != slice'
    lbl.set_gather_mode('auto')
    auto = lbl.extinction(t, d, z, add=add).clone()
    check('auto', auto)
    again = lbl.extinction(t, d, z, add=add)
    check('auto (second call)', again)
    assert torch.equal(again, auto), 'auto: two runs differ'
    for env in ({'PB_TILE_SPLIT': '0'}, {'PB_TILE_GLOBAL': '0'}):
        os.environ.update(env)
        p2 = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext, vg['cutoff'],
                     ethresh, max_layers=nlayers)
        alt = p2.extinction(t, d, z, add=add)
        for k in env:
            os.environ.pop(k)
        check(str(env), alt)
        rel = ((alt - auto).abs() / auto.abs().clamp_min(1e-300)).max().item()
        assert torch.equal(alt == 0, auto == 0) and rel <= 1e-12, f'{env}: {rel:.2e}'
        info['tiles'] = info['tiles'] or not torch.equal(alt, auto)
        p2.close()
    lbl.close()
    return info


def partition_case(eng, rng):
    from pyratbay_amd import tli
    dbs = []
    for _ in range(int(rng.integers(1, 4))):
        nt = int(rng.integers(2, 400))
        t = np.cumsum(rng.uniform(0.5, 40.0, nt)) + rng.uniform(1.0, 200.0)
        pf = np.cumsum(rng.uniform(0.0, 50.0, (int(rng.integers(1, 6)), nt)), axis=1) + 1.0
        dbs.append(dict(temperatures=t, partition=pf, isotopes=['x'] * pf.shape[0]))
    lo = max(d['temperatures'][0] for d in dbs)
    hi = min(d['temperatures'][-1] for d in dbs)
    if hi <= lo:
        return
    temps = rng.uniform(lo, hi, (int(rng.integers(1, 9)), int(rng.integers(1, 90))))
    for d in dbs:                                     # some temperatures exactly on nodes
        node = d['temperatures'][(d['temperatures'] >= lo) & (d['temperatures'] <= hi)]
        if len(node):
            temps.flat[rng.integers(0, temps.size)] = node[rng.integers(0, len(node))]
    pt = eng.PartitionTable(dbs)
    got = pt.evaluate(eng.dev(temps)).cpu().numpy()
    want = tli.iso_partition(dbs, temps.ravel()).reshape(got.shape)
    assert np.array_equal(got, want), 'partition table: device != host'


def ordered_case(eng, rng):
    """The retrieval batches with their columns in depth order (pb_transit_spectrum_ordered,
    pb_emission_flux_ordered through TableSpectrum.eval_bands) against grid order: band fluxes
    bit for bit; the transit spectra of a random permutation against the oracle."""
    import torch
    from oracle import oracle as orc
    from pyratbay_amd import synth
    nspec, ntemp = int(rng.integers(1, 6)), int(rng.integers(2, 9))
    L, W, nw = int(rng.integers(2, 120)), int(rng.integers(64, 3000)), int(rng.integers(1, 9))
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    lo = int(rng.integers(0, W // 2))
    bands = [(lo, np.ones(W - 1 - lo), 1.0)]
    pb = eng.PassBands(g['wn'], bands)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    radius = radius0[None] * (1 + 0.01 * rng.uniform(-1, 1, (nw, 1)))
    args = [eng.dev(x) for x in (temps, dens)]
    for rt in ('transit', 'emission'):
        out = []
        for order in (None, 'auto', rng.permutation(W)):
            model = eng.TableSpectrum(etable, ttable, g['wn'], radius0, 8.8e10, rt_path=rt,
                                      column_order=order)
            out.append(model.eval_bands(*args, pb, radius=eng.dev(radius)).clone())
        assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2]), f'{rt}: order changes bits'
    # one walker's spectrum through the ordered transit kernel against the oracle
    if L >= 2:
        ec = eng.interp_ec_batch(eng.dev(etable), eng.dev(ttable), args[0][:1], args[1][:1])
        rad = eng.dev(radius[:1])
        order = torch.as_tensor(rng.permutation(W), device='cuda')
        got = eng.transit_spectrum_ordered(ec[:, :, order].contiguous(),
                                           eng.transit_path_device(rad, 0), rad,
                                           order.to(torch.int32), 8.8e10, 0, L, 10.0)
        wd, wi = orc.optical_depth_transit(ec[0].cpu().numpy(), radius[0], 0, L, 10.0)
        ws = orc.transmission(wd, radius[0], 8.8e10, wi, 0)
        np.testing.assert_allclose(got[0].cpu().numpy(), ws, rtol=1e-12)


def predicted_case(eng, rng, seed):
    """`resolution` mode with the run plan predicted from the last read-back
    (LBLSpectrum(predict_runs=True)) against the default (synchronous) form over a random sequence
    of atmospheres on ONE pair of models: bit for bit while the atmosphere repeats, 1e-12 with the
    same zero pattern when it changes (the layers the plan does not fit take the direct gather)."""
    import torch
    from pyratbay_amd import synth
    nl = int(rng.integers(2, 12))
    case = synth.lbl_case(int(rng.integers(600, 2500)), nl, int(rng.integers(500, 6000)),
                          wnosamp=int(rng.choice([12, 24, 36])), nlor=14, ndop=7, extent=60.0,
                          cutoff=3.0, niso=2, seed=seed, resolution=float(rng.choice([3e4, 6e4, 1e5])))
    atm, iso = case['atm'], case['iso']
    ref = eng.LBLSpectrum(case, timestamps=False)
    mod = eng.LBLSpectrum(case, voigt=ref.voigt, lines=ref.lines, timestamps=False,
                          predict_runs=True)
    state = (atm['temp'].copy(), atm['dens'].copy(), iso['isoz'].copy())
    for step in range(int(rng.integers(6, 16))):
        changed = step > 0 and rng.random() < 0.4
        if changed:
            perm = rng.permutation(nl) if rng.random() < 0.5 else np.arange(nl)
            state = (atm['temp'][perm] * rng.uniform(0.7, 1.4), atm['dens'][perm] *
                     10.0**rng.uniform(-1.5, 1.0), iso['isoz'][:, perm].copy())
            ref.set_atmosphere(*state)
            mod.set_atmosphere(*state)
        want, want_ec = ref.run().clone(), ref.ec.clone()
        got = mod.run()
        e, w = mod.ec.cpu().numpy(), want_ec.cpu().numpy()
        assert np.array_equal(e == 0, w == 0), f'step {step}: zero pattern'
        np.testing.assert_allclose(e, w, rtol=1e-12, err_msg=f'step {step}')
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-12)
        if rng.random() < 0.5:
            torch.cuda.synchronize()                  # (lets the read-back land before the next call)
    spec, sync, missed = mod.lbl.dyn_stats()
    return spec, missed


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 41000
    from oracle import ref
    from pyratbay_amd import engine as eng
    eng.require_gpu()
    assert ref.available(), 'fuzz_r4 needs the compiled reference (oracle/_ref)'
    fails, nwave_layers, ntiles, nband, npred, nmiss = 0, 0, 0, 0, 0, 0
    for i in range(count):
        seed = seed0 + i
        rng = np.random.default_rng(seed)
        try:
            if i % 4 == 3:
                for _ in range(20):
                    partition_case(eng, rng)
                for _ in range(3):
                    ordered_case(eng, rng)
                from pyratbay_amd import _capi
                if _capi.experiments():                   # (predicted run plans: libpbhip_exp.so only)
                    sp, mi = predicted_case(eng, rng, seed)
                    npred += sp
                    nmiss += mi
            else:
                info = ext_case(eng, ref, rng, seed)
                nwave_layers += info['wave']
                ntiles += int(info['tiles'])
                nband += int(info['banded'])
        except Exception:                                       # noqa: BLE001
            fails += 1
            print(f'FAIL seed {seed}')
            traceback.print_exc()
            if fails >= 5:
                break
    print(f'fuzz_r4: {count} cases from seed {seed0}: {fails} failures; wave-kernel layers '
          f'{nwave_layers}, cases where the per-tile dispatch changed the launch {ntiles}, '
          f'band-structured lists {nband}; `resolution` calls planned from a prediction {npred}, '
          f'contradicted read-backs {nmiss}')
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
