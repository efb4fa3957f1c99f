"""BASELINE.json's configurations on the HIP path, inside `pytest -m gpu`.

* C4's structure at reduced size: four line-carrying species, several isotopes each, one
  extinction row per species (add=0: per-row kmax / ethresh, _extcoeff.c:203-226,265-272) and
  the summed form (add=1, pyrat/extinction.py:170-213), every gather kernel, against the oracle.
* C2, C3 and C4 at FULL size: size-independent properties of the HIP path (wavenumber shards
  concatenate bit for bit, two runs are bitwise equal, the LDS-staged and the global gather
  agree to 1e-12) plus oracle parity on sampled layers (the oracle's C restatement of
  _extcoeff.extinction on one host core; a few seconds per layer).

* C5 at FULL size: one 64-walker batch of the retrieval inner loop -- run-to-run bitwise equal,
  independent of how the walkers are chunked, out-of-range temperatures rejected, the one-walker
  eval() chain and the oracle chain on sampled walkers.
* C1 (the reference's CPU-runnable tutorial shape: 4 501 wavenumbers x 51 layers, one species):
  the whole path, every layer of ec and the transit / emission spectrum against the oracle.

Tolerance as in test_gpu_extinction.py: rtol 1e-10 against the oracle on every non-zero
sample and an identical zero pattern."""
import time

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-10

C4_SPECIES = ('H2', 'He', 'H2O', 'CO', 'CO2', 'CH4')
C4_VMR = (0.85, 0.149, 4e-4, 5e-4, 1e-7, 1e-4)
C4_LINES = ('H2O', 'CO', 'CO2', 'CH4')


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


def host(t):
    return t.cpu().numpy()


def oracle_rows(orc, case, vt, profile, layer, add, ethresh=1e-30):
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    rows = 1 if add else int(iso['isoiext'].max()) + 1
    want = np.zeros((rows, g['nwave']))
    orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                   g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                   atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                   iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                   ln['gf'], ln['lid'], vg['cutoff'], ethresh, atm['temp'][layer], 0,
                   int(add), 0)
    return want


def plan(eng, case, max_layers):
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'])
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], len(iso['isomass']),
                      g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], case['ethresh'], max_layers=max_layers)
    return vt, ll, lbl


def check(got, want, what):
    assert np.array_equal(got == 0, want == 0), f'{what}: zero pattern differs'
    nz = want != 0
    worst = float(np.max(np.abs(got[nz] / want[nz] - 1))) if nz.any() else 0.0
    np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=what)
    return worst


# ---------------------------------------------------------------------------
# C4's structure, reduced size
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('gather', cases.gathers('global', 'staged', 'resident', 'scatter',
                                                  'rounds', 'auto'))
@pytest.mark.parametrize('ethresh', [1e-30, 1e-4])
def test_four_species_vs_oracle(eng, orc, gather, ethresh):
    from pyratbay_amd import synth
    case = synth.lbl_case(6001, 7, 9000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=3, seed=31, species=C4_SPECIES, vmr=C4_VMR,
                          line_species=C4_LINES)
    iso, atm = case['iso'], case['atm']
    assert list(iso['isoiext']) == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3
    vt, ll, lbl = plan(eng, case, 7)
    lbl.set_gather_mode(gather)
    lbl.set_ethresh(ethresh)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    profile = vt.flat()
    worst = 0.0
    for add in (False, True):
        ext = host(lbl.extinction(t, d, z, add=add))
        assert ext.shape == (7, 1 if add else 4, case['grid']['nwave'])
        for layer in range(7):
            want = oracle_rows(orc, case, vt, profile, layer, add, ethresh)
            worst = max(worst, check(ext[layer], want, f'{gather} add={add} layer {layer}'))
        if not add:
            # every species row is populated and they differ (per-row kmax and density)
            assert all(np.any(ext[:, r] != 0) for r in range(4))
            # per-row maxima: the strongest in-range line of each species
            _, kmax = lbl.last_state(7, 4)
            assert np.all(kmax > 0) and len(np.unique(kmax[3])) == 4
    print(f'4 species {gather} ethresh={ethresh}: max rel err vs oracle = {worst:.2e}')


def test_four_species_shards_and_rows(eng, orc):
    """add=0 rows of a 4-species list: shards concatenate, a species switched off
    (isoiext = -1) leaves the other rows untouched."""
    from pyratbay_amd import synth
    case = synth.lbl_case(5001, 4, 12000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=2, seed=32, species=C4_SPECIES, vmr=C4_VMR,
                          line_species=C4_LINES)
    iso, atm = case['iso'], case['atm']
    vt, ll, lbl = plan(eng, case, 4)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    for gather in cases.live('staged', 'global', 'rounds'):
        lbl.set_isoiext(iso['isoiext'])
        lbl.set_gather_mode(gather)
        full = host(lbl.extinction(t, d, z, add=False))
        parts = [host(lbl.extinction(t, d, z, add=False, wbegin=a, wcount=b - a))
                 for a, b in ((0, 1111), (1111, 4096), (4096, 5001))]
        assert np.array_equal(np.concatenate(parts, axis=2), full), gather
        off = iso['isoiext'].copy()
        off[2:4] = -1                                      # CO off
        lbl.set_isoiext(off)
        part = host(lbl.extinction(t, d, z, add=False))
        assert np.all(part[:, 1] == 0)
        assert np.array_equal(part[:, [0, 2, 3]], full[:, [0, 2, 3]]), gather


# ---------------------------------------------------------------------------
# full-size configurations
# ---------------------------------------------------------------------------
FULL = {
    # BASELINE.json configs[1]
    'c2': dict(args=(100001, 80, 100000), kw=dict(wnstep=0.05, niso=1), layers=(0, 41, 79)),
    # north_star's speed-up configuration: the C2 grid with a 1e6-line list
    'c2-1e6': dict(args=(100001, 80, 1000000), kw=dict(wnstep=0.05, niso=1), layers=(41,)),
    # configs[2]: 1e6 wavenumbers, 1e6-line 4-isotope list
    'c3': dict(args=(1000001, 80, 1000000), kw=dict(wnstep=0.005, niso=4), layers=(3, 76)),
    # C3 with a band-structured list (synth.band_positions: band heads at 300 x the background's
    # line density, duplicated positions): the reference's own LBL tests run on HITRAN, whose
    # density contrasts the uniform lists above do not have
    'c3-bands': dict(args=(1000001, 80, 1000000), kw=dict(wnstep=0.005, niso=4, bands=True),
                     layers=(3, 76)),
    # configs[3]: 1e6 wavenumbers x 120 layers, 4 species x 1e6 lines (single-GPU form)
    'c4': dict(args=(1000001, 120, 1000000),
               kw=dict(wnstep=0.005, niso=4, species=C4_SPECIES, vmr=C4_VMR,
                       line_species=C4_LINES), layers=(10, 112)),
}


@pytest.mark.parametrize('name', ['c2', 'c2-1e6', 'c3', 'c3-bands', 'c4'])
def test_full_size_config(eng, orc, name, monkeypatch):
    import torch
    from pyratbay_amd import synth
    cfg = FULL[name]
    t0 = time.time()
    case = synth.lbl_case(*cfg['args'], seed=42, **cfg['kw'])
    g, atm, iso = case['grid'], case['atm'], case['iso']
    nl, nw = atm['nlayers'], g['nwave']
    vt, ll, lbl = plan(eng, case, nl)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    print(f'{name}: W={nw} L={nl} lines={ll.nlines} groups={ll.ngroups} '
          f'isotopes={len(iso["isomass"])} set-up {time.time() - t0:.1f} s')

    # (1) default kernel choice, twice: bitwise equal
    full = lbl.extinction(t, d, z, add=True)
    kernel = lbl.last_gather_kernel
    again = lbl.extinction(t, d, z, add=True)
    assert torch.equal(full, again), 'two runs differ'
    del again
    assert kernel.endswith('k_ext_staged'), kernel

    # (2) wavenumber shards (8 uneven ones, as an 8-GPU run would cut them) with the same
    # kernel and the same phase split forced (the split of a launch is a performance choice
    # that changes the association of the per-sample sums): bit for bit
    # the full-grid result
    lbl.set_gather_mode('staged')
    monkeypatch.setenv('PB_STAGE_SPLIT', '1')
    ref_staged = lbl.extinction(t, d, z, add=True)
    bounds = np.linspace(0, nw, 9).astype(int)
    bounds[3] += 1717                                       # uneven, not tile-aligned
    for a, b in zip(bounds[:-1], bounds[1:]):
        part = lbl.extinction(t, d, z, add=True, wbegin=int(a), wcount=int(b - a))
        assert torch.equal(part, ref_staged[:, :, a:b]), f'shard [{a},{b}) differs'
        del part

    # (2b) the line list walked in chunks under a record budget of 1/3.4 of the whole list's
    # records (out-of-core line lists: pb_lbl_set_record_budget): bit for bit the one-call result
    lbl.set_record_budget(int(ll.ngroups * nl * 16 / 3.4))
    t1 = time.time()
    cut = lbl.extinction(t, d, z, add=True)
    torch.cuda.synchronize()
    nchunks = lbl.last_chunks
    assert nchunks >= 4, nchunks
    assert torch.equal(cut, ref_staged), 'chunked line list differs from the one-call result'
    del cut
    lbl.set_record_budget(96 << 30)
    print(f'{name}: {nchunks} chunks of the line list equal the one-call result '
          f'({time.time() - t1:.2f} s)')

    monkeypatch.delenv('PB_STAGE_SPLIT')
    # (3) staged vs global gather: same terms, different association
    lbl.set_gather_mode('global')
    glob = lbl.extinction(t, d, z, add=True)
    assert lbl.last_gather_kernel.endswith('k_ext_resample')
    assert torch.equal(glob == 0, ref_staged == 0)
    rel = ((glob - ref_staged).abs() / ref_staged.abs().clamp_min(1e-300)).max().item()
    assert rel <= 1e-12, rel
    assert torch.equal(full == 0, ref_staged == 0)
    rel_default = ((full - ref_staged).abs() / ref_staged.abs().clamp_min(1e-300)).max().item()
    assert rel_default <= 1e-12, rel_default
    del glob

    # (4) oracle parity on sampled layers, full size
    profile = vt.flat()
    ec = host(full[list(cfg['layers'])])
    worst = 0.0
    for i, layer in enumerate(cfg['layers']):
        t1 = time.time()
        want = oracle_rows(orc, case, vt, profile, layer, True)
        worst = max(worst, check(ec[i], want, f'{name} layer {layer}'))
        print(f'{name}: layer {layer} oracle {time.time() - t1:.1f} s')
    print(f'{name}: kernel {kernel}; shards exact; staged vs global {rel:.1e}; '
          f'max rel err vs oracle on layers {cfg["layers"]} = {worst:.2e}')

    if name == 'c4':
        # one row per species on a few layers: per-row maxima, rows sum to the add=1 result
        sub = [10, 60, 112]
        lbl.set_gather_mode('auto')
        ts, ds, zs = t[sub].contiguous(), d[sub].contiguous(), z[:, sub].contiguous()
        rows = lbl.extinction(ts, ds, zs, add=False)
        assert rows.shape == (3, 4, nw)
        dens_rows = torch.stack([d[sub][:, 2 + r] for r in range(4)], dim=1)   # [3, 4]
        total = (rows * dens_rows[:, :, None]).sum(dim=1)
        want = full[sub][:, 0]
        relr = ((total - want).abs() / want.abs().clamp_min(1e-300)).max().item()
        assert relr <= 1e-12, relr
        want0 = oracle_rows(orc, case, vt, profile, 10, False)
        check(host(rows[0]), want0, 'c4 add=0 layer 10')


# ---------------------------------------------------------------------------
# C1: the reference's CPU-runnable tutorial shape (W ~ 3.2-5 k, L ~ 40-51, one species) through
# the whole path, every layer against the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('rt_path', ['transit', 'emission'])
def test_c1_tutorial_shape(eng, orc, rt_path):
    from pyratbay_amd import synth
    case = synth.lbl_case(4501, 51, 6000, wnstep=0.2, nlor=40, ndop=20, seed=5)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    nl, nw = atm['nlayers'], g['nwave']
    assert (nl, nw) == (51, 4501)
    model = eng.LBLSpectrum(case, rt_path=rt_path)
    spectrum = host(model.run())
    profile = model.voigt.flat()
    ec = np.zeros((nl, nw))
    for layer in range(nl):
        ec[layer] = oracle_rows(orc, case, model.voigt, profile, layer, True, case['ethresh'])[0]
    check(host(model.ec)[:, 0], ec, f'c1 {rt_path} ec')
    if rt_path == 'transit':
        depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, nl, case['maxdepth'])
        want = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    else:
        depth = np.zeros((nl, nw))
        ideep = np.full(nw, nl - 1, np.int32)
        orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(atm['radius']),
                                         case['maxdepth'], 0, nl)
        inten = orc.intensity(depth, ideep, orc.blackbody_wn_2D(g['wn'], atm['temp']),
                              host(model.mu), 0)
        want = np.sum(inten * host(model.weights)[:, None], axis=0)
    assert np.array_equal(host(model.ideep), ideep)
    np.testing.assert_allclose(spectrum, want, rtol=RTOL)
    assert want.max() / want.min() > 1.0005


# ---------------------------------------------------------------------------
# C5 at FULL size: one batch of 64 walkers of the retrieval inner loop (1e5 wavenumbers x 80
# layers, 4 species x 10 table temperatures = 2.56 GB of cross sections, per-walker radius)
# ---------------------------------------------------------------------------
def test_full_size_c5_batch(eng, orc):
    import torch
    from tools import bench_c5
    inp = bench_c5.inputs()
    g, atm = inp['grid'], inp['atm']
    nl, nw = atm['nlayers'], g['nwave']
    assert (nl, nw, inp['etable'].shape[:2]) == (80, 100001, (4, 10))
    model = eng.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'], atm['rstar'])
    bands = eng.PassBands(g['wn'], inp['bands'])
    temps, dens, radius = bench_c5.walkers(inp, 64, 700)
    temps[17, 40] = 3000.5                               # above the table: rejected
    td, dd, rd = eng.dev(temps), eng.dev(dens), eng.dev(radius)
    flux = model.eval_bands(td, dd, bands, radius=rd, chunk=64)
    again = model.eval_bands(td, dd, bands, radius=rd, chunk=64)
    assert torch.equal(flux, again), 'two runs differ'
    got = host(flux)
    assert got.shape == (64, 24)
    assert np.all(np.isinf(got[17])) and np.all(got[17] > 0)
    ok = [w for w in range(64) if w != 17]
    assert np.all(np.isfinite(got[ok]))
    # a walker's result does not depend on the batch it travels in (same kernels: chunks of 32)
    halves = torch.cat([model.eval_bands(td[a:b], dd[a:b], bands, radius=rd[a:b], chunk=32)
                        for a, b in ((0, 32), (32, 64))])
    assert torch.equal(halves[ok], flux[ok])
    # the one-walker eval() chain (reference products and sums) agrees to rounding
    for w in (0, 45):
        model.set_radius(radius[w])
        one = host(bands.integrate_batch(model.eval(temps[w], dens[w]).view(1, -1)))[0]
        np.testing.assert_allclose(got[w], one, rtol=1e-12)
    # oracle: interp_ec -> transit_path -> optical depth -> transmission -> trapezoid
    for w in (3, 60):
        ec = np.zeros((nl, nw))
        orc.interp_ec(ec, inp['etable'], inp['ttable'], temps[w], dens[w], 0, nl)
        depth, ideep = orc.optical_depth_transit(ec, radius[w], 0, nl, 10.0)
        spec = orc.transmission(depth, radius[w], atm['rstar'], ideep, 0)
        want = [np.trapezoid(spec[s:s + len(r)] * r, g['wn'][s:s + len(r)]) * h
                for s, r, h in inp['bands']]
        np.testing.assert_allclose(got[w], want, rtol=RTOL)


# ---------------------------------------------------------------------------
# C3 as BASELINE.json states it: "1e6 wavenumbers x 80 layers, 1e6-line list, emission +
# _blackbody path" -- the whole run() at full size; the RT half (plane_parallel_optical_depth ->
# blackbody_wn_2D -> intensity -> quadrature sum: _trapezoid.c:175-213, 304-341,
# _blackbody.c:35-75, pyrat/spectrum.py:366-377) against the oracle on EVERY column
# ---------------------------------------------------------------------------
def test_full_size_c3_emission(eng, orc):
    import torch
    from pyratbay_amd import synth
    cfg = FULL['c3']
    t0 = time.time()
    case = synth.lbl_case(*cfg['args'], seed=42, **cfg['kw'])
    g, atm = case['grid'], case['atm']
    nl, nw = atm['nlayers'], g['nwave']
    assert (nl, nw) == (80, 1000001)
    model = eng.LBLSpectrum(case, rt_path='emission')
    spectrum = model.run()
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    again = model.run()
    assert torch.equal(again, spectrum), 'two runs differ'
    assert model.lbl.last_gather_kernel.endswith('k_ext_staged')
    assert model.depth.shape == (nl, nw) and model.ideep.shape == (nw,)
    ec = host(model.ec)[:, 0]
    # extinction: oracle parity on one sampled layer (test_full_size_config[c3] covers more)
    profile = model.voigt.flat()
    layer = 44
    want_row = oracle_rows(orc, case, model.voigt, profile, layer, True, case['ethresh'])
    worst_ec = check(ec[layer:layer + 1], want_row, f'c3 emission ec layer {layer}')
    del profile
    # RT stages from the HIP ec, every column, on the host
    t1 = time.time()
    depth = np.zeros((nl, nw))
    ideep = np.full(nw, nl - 1, np.int32)
    orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(atm['radius']),
                                     case['maxdepth'], 0, nl)
    B = orc.blackbody_wn_2D(g['wn'], atm['temp'])
    mu, weights = host(model.mu), host(model.weights)
    inten = orc.intensity(depth, ideep, B, mu, 0)
    want = np.sum(inten * weights[:, None], axis=0)
    t_orc = time.time() - t1
    assert np.array_equal(host(model.ideep), ideep), 'ideep differs'
    got_depth = host(model.depth)
    np.testing.assert_allclose(got_depth, depth, rtol=1e-12, atol=0)
    got = host(spectrum)
    assert np.all(np.isfinite(got)) and np.all(got > 0)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    # the columns stop at very different layers (line cores high up, windows at the bottom) and
    # the last columns of the 8e7-element arrays are as right as the first (64-bit indexing)
    assert ideep.max() - ideep.min() >= 10 and ideep.max() <= nl - 1
    np.testing.assert_allclose(got[-1000:], want[-1000:], rtol=RTOL)
    print(f'c3 emission: W={nw} L={nl}; set-up + first run {t_gpu:.1f} s; oracle RT on all '
          f'columns {t_orc:.1f} s; ec layer {layer} {worst_ec:.1e}; spectrum max rel err '
          f'{np.max(np.abs(got / want - 1)):.2e}; ideep {ideep.min()}..{ideep.max()} exact')


def test_full_size_c5_emission_batch(eng, orc):
    """C5's batch in emission geometry at full size (1e5 wavenumbers x 80 layers x 64 walkers,
    `bench.py --workload c5-emission`): run-to-run bitwise equal, independent of the chunking
    and of the column order, rejects, and sampled walkers against the oracle chain interp_ec ->
    plane-parallel depth -> Planck -> intensity -> quadrature -> band trapezoids."""
    import torch
    from tools import bench_c5
    inp = bench_c5.inputs()
    g, atm = inp['grid'], inp['atm']
    nl, nw = atm['nlayers'], g['nwave']
    model = eng.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'], atm['rstar'],
                              rt_path='emission')
    bands = eng.PassBands(g['wn'], inp['bands'])
    temps, dens, radius = bench_c5.walkers(inp, 64, 700)
    temps[9, 11] = 299.0                                 # below the table: rejected
    td, dd, rd = eng.dev(temps), eng.dev(dens), eng.dev(radius)
    flux = model.eval_bands(td, dd, bands, radius=rd, chunk=64)
    assert model.column_order is not None                # (ordered by the first walker's depth)
    again = model.eval_bands(td, dd, bands, radius=rd, chunk=64)
    assert torch.equal(flux, again), 'two runs differ'
    got = host(flux)
    assert got.shape == (64, 24) and np.all(np.isinf(got[9])) and np.all(got[9] > 0)
    ok = [w for w in range(64) if w != 9]
    assert np.all(np.isfinite(got[ok]))
    halves = torch.cat([model.eval_bands(td[a:b], dd[a:b], bands, radius=rd[a:b], chunk=32)
                        for a, b in ((0, 32), (32, 64))])
    assert torch.equal(halves[ok], flux[ok])
    del model
    torch.cuda.empty_cache()
    grid = eng.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'], atm['rstar'],
                             rt_path='emission', column_order=None)
    assert torch.equal(grid.eval_bands(td, dd, bands, radius=rd, chunk=64)[ok], flux[ok])
    mu, weights = eng.default_quadrature()
    for w in (2, 33, 63):
        ec = np.zeros((nl, nw))
        orc.interp_ec(ec, inp['etable'], inp['ttable'], temps[w], dens[w], 0, nl)
        depth = np.zeros((nl, nw))
        ideep = np.zeros(nw, np.int32)
        orc.plane_parallel_optical_depth(depth, ideep, ec, -np.diff(radius[w]), 10.0, 0, nl)
        inten = orc.intensity(depth, ideep, orc.blackbody_wn_2D(g['wn'], temps[w]), mu, 0)
        spec = np.sum(inten * weights[:, None], axis=0)
        want = [np.trapezoid(spec[s:s + len(r)] * r, g['wn'][s:s + len(r)]) * h
                for s, r, h in inp['bands']]
        np.testing.assert_allclose(got[w], want, rtol=RTOL)
