"""Walker-batched retrieval path (BASELINE config 5) and the fused transit column kernel, through
the C ABI, against the oracle.  Reference inner loop: pyratbay/pyrat/pyrat_obj.py:225-385 ->
opacity/line_sampling.py:394-463, atmosphere/atmosphere.py:782-802, opacity/optic_depth.py:103-112,
spectrum/radiative_transfer.py:57-71, spectrum/spec_tools.py:193-233.  Tolerance 1e-12."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


def host(t):
    return t.cpu().numpy()


def hydro_radius(rng, nlayers, scale):
    """A smooth, strictly decreasing radius profile, different for every walker."""
    base = np.linspace(8.0e9, 7.0e9, nlayers)
    return base * (1.0 + scale * rng.uniform(-1, 1)) + np.linspace(0, 1, nlayers) * 2e7 * rng.uniform(-1, 1)


def test_transit_path_device(eng, orc):
    rng = np.random.default_rng(2)
    nw, L = 5, 37
    radius = np.array([hydro_radius(rng, L, 0.02) for _ in range(nw)])
    for itop in (0, 3):
        got = host(eng.transit_path_device(eng.dev(radius), itop))
        assert np.all(np.isfinite(got))
        for w in range(nw):
            want = eng.pack_raypath(eng.transit_path(radius[w], itop), itop)
            want_o = eng.pack_raypath(orc.transit_path(radius[w], itop), itop)
            assert np.array_equal(want, want_o)
            # the host form squares with libm's pow like the reference, the kernel with one
            # multiply: equal bits wherever pow(x, 2) == x*x (99.9 % of radii), else one ulp of
            # the square apart
            exact = all(x**2 == x * x for x in radius[w, itop:].tolist())
            if exact:
                assert np.array_equal(got[w], want), (itop, w)
            np.testing.assert_allclose(got[w], want, rtol=1e-11)


@pytest.mark.parametrize('L,W', [(9, 1500), (9, 1501), (8, 1501), (8, 5), (8, 3), (10, 258)])
def test_interp_ec_batch_vs_single_and_oracle(eng, orc, L, W):
    """The batched interpolation against the per-walker kernel (bit for bit) and the oracle.
    (L, W): even rows -- pairs of samples from sample 0; L * W odd -- the one-sample kernel; odd W
    with even L -- every other row starts at an odd element, its first sample is taken alone and
    the pairs follow (16-byte aligned accesses); W = 5 / 3 -- the smallest grids with / without
    pairs; W = 258 -- a pair in the second workgroup."""
    rng = np.random.default_rng(4)
    nmol, ntemp, nw = 4, 7, 37
    ttable = np.linspace(300.0, 3000.0, ntemp)
    etable = 10.0**rng.uniform(-30, -20, (nmol, ntemp, L, W))
    temps = rng.uniform(300.0, 3000.0, (nw, L))
    temps[0] = ttable[rng.integers(0, ntemp, L)]          # exactly on nodes
    temps[1] = 300.0
    temps[2] = 3000.0
    temps[3, :] = np.linspace(640, 660, L)                # one bracket
    dens = 10.0**rng.uniform(8, 18, (nw, L, nmol))
    et, tt = eng.dev(etable), eng.dev(ttable)
    got = host(eng.interp_ec_batch(et, tt, eng.dev(temps), eng.dev(dens)))
    import torch
    for w in range(nw):
        one = torch.full((L, W), 3.3, dtype=torch.float64, device='cuda')
        eng.interp_ec(one, et, tt, eng.dev(temps[w]), eng.dev(dens[w]), 0, L, assign=True)
        assert np.array_equal(got[w], host(one)), w         # same arithmetic, bit for bit
    for w in (0, 1, 2, 3, 17, 36):
        want = np.zeros((L, W))
        orc.interp_ec(want, etable, ttable, temps[w], dens[w], 0, L)
        np.testing.assert_allclose(got[w], want, rtol=RTOL)
    # 6 species (the 8-register instantiation)
    etable6 = 10.0**rng.uniform(-30, -20, (6, ntemp, L, 300))
    dens6 = 10.0**rng.uniform(8, 18, (5, L, 6))
    got6 = host(eng.interp_ec_batch(eng.dev(etable6), tt, eng.dev(temps[:5]), eng.dev(dens6)))
    for w in range(5):
        want = np.zeros((L, 300))
        orc.interp_ec(want, etable6, ttable, temps[w], dens6[w], 0, L)
        np.testing.assert_allclose(got6[w], want, rtol=RTOL)


@pytest.mark.parametrize('rows', [None, 16, 'mfma'])
@pytest.mark.parametrize('itop,ibottom,maxdepth', [(0, None, 10.0), (2, None, 10.0),
                                                   (0, 19, 10.0), (1, None, np.inf)])
def test_transit_spectrum_batch_vs_oracle(eng, orc, itop, ibottom, maxdepth, rows, monkeypatch):
    """The spectrum-only call of a batch runs on the matrix cores by default (k_transit_mfma:
    rows='mfma'); with PB_TRANSIT_MFMA=0 it runs the vector kernels: rows=16 forces the block
    size of large launches at this small size, which selects k_transit_pair (two columns per
    thread, fused multiply-adds); rows=None the one-column kernel with the reference's products
    and sums, which is also what the call that returns depth always runs."""
    if rows != 'mfma':
        monkeypatch.setenv('PB_TRANSIT_MFMA', '0')
    if rows == 16:
        monkeypatch.setenv('PB_TRANSIT_ROWS', str(rows))
    rng = np.random.default_rng(7)
    c = cases.column_case(seed=9, nlayers=24, nwave=700)
    L, W, nw = c['nlayers'], c['nwave'], 6
    ibottom = L if ibottom is None else ibottom
    ecs = np.array([c['ec'] * 10.0**rng.uniform(-1, 1) for _ in range(nw)])
    radius = np.array([np.sort(c['radius'] * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    rad_d = eng.dev(radius)
    path = eng.transit_path_device(rad_d, itop)
    spec, depth, ideep = eng.transit_spectrum_batch(eng.dev(ecs), path, rad_d, c['rstar'], itop,
                                                    ibottom, maxdepth, want_depth=True)
    only = eng.transit_spectrum_batch(eng.dev(ecs), path, rad_d, c['rstar'], itop, ibottom,
                                      maxdepth)
    if rows:
        np.testing.assert_allclose(host(only), host(spec), rtol=1e-13)
    else:
        assert np.array_equal(host(only), host(spec))
    for w in range(nw):
        wd, wi = orc.optical_depth_transit(ecs[w], radius[w], itop, ibottom, maxdepth)
        ws = orc.transmission(wd, radius[w], c['rstar'], wi, itop)
        assert np.array_equal(host(ideep[w]), wi)
        np.testing.assert_allclose(host(depth[w]), wd, rtol=RTOL, atol=0)
        np.testing.assert_allclose(host(spec[w]), ws, rtol=RTOL)
        np.testing.assert_allclose(host(only[w]), ws, rtol=RTOL)


@pytest.mark.parametrize('nlayers', [2, 9, 16, 17, 40, 64, 81, 100, 128])
def test_transit_ordered_columns(eng, orc, monkeypatch, nlayers):
    """pb_transit_spectrum_ordered (row tile by row tile, a wavefront stops where its 32 columns
    have all crossed maxdepth): for the columns in a random order and in the order of their first
    crossing, every instantiation (1 ... 8 row tiles), ragged widths, top layers, bottoms and
    maximum depths that stop early / never -- the spectra in grid order, bit for bit those of
    pb_transit_spectrum_batch on the unpermuted ec (the row-tile kernel in grid order; in the
    experiments build also the layers-outer kernel, PB_TRANSIT_MFMA=4), 1e-13 of the vector
    kernels, and of the oracle."""
    import torch
    rng = np.random.default_rng(40 + nlayers)
    W = int(rng.choice([2, 31, 33, 700, 1601]))
    nw = 3
    c = cases.column_case(seed=60 + nlayers, nlayers=nlayers, nwave=W)
    # columns that cross at very different rows (a line core beside a window)
    ecs = np.array([c['ec'] * 10.0**rng.uniform(-3, 3, (1, W)) * 10.0**rng.uniform(-1, 1)
                    for _ in range(nw)])
    radius = np.array([np.sort(c['radius'] * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    rad_d, ec_d = eng.dev(radius), eng.dev(ecs)
    for itop, ibottom, maxdepth in ((0, nlayers, 10.0), (nlayers // 3, nlayers, 0.3),
                                    (0, max(2, nlayers - 1), np.inf)):
        if min(ibottom, nlayers) - itop < 2:
            continue
        path = eng.transit_path_device(rad_d, itop)
        ref = eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, maxdepth)
        if cases.EXPERIMENTS:
            # (the layers-outer matrix kernel of round 3, experiments build only: same bits)
            monkeypatch.setenv('PB_TRANSIT_MFMA', '4')
            outer = eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom,
                                               maxdepth)
            monkeypatch.delenv('PB_TRANSIT_MFMA')
            assert torch.equal(outer, ref)
        _, _, ideep = eng.transit_spectrum_batch(ec_d[:1], path[:1], rad_d[:1], c['rstar'], itop,
                                                 ibottom, maxdepth, want_depth=True)
        by_depth = torch.sort(ideep[0], stable=True).indices
        for order in (torch.as_tensor(rng.permutation(W), device='cuda'), by_depth):
            got = eng.transit_spectrum_ordered(ec_d[:, :, order].contiguous(), path, rad_d,
                                               order.to(torch.int32), c['rstar'], itop, ibottom,
                                               maxdepth)
            assert torch.equal(got, ref), (itop, ibottom, maxdepth)
        monkeypatch.setenv('PB_TRANSIT_MFMA', '0')
        vec = eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, maxdepth)
        monkeypatch.delenv('PB_TRANSIT_MFMA')
        np.testing.assert_allclose(host(ref), host(vec), rtol=1e-13)
    wd, wi = orc.optical_depth_transit(ecs[1], radius[1], 0, nlayers, 10.0)
    ws = orc.transmission(wd, radius[1], c['rstar'], wi, 0)
    path = eng.transit_path_device(rad_d, 0)
    got = eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], 0, nlayers, 10.0)
    np.testing.assert_allclose(host(got[1]), ws, rtol=RTOL)


@pytest.mark.parametrize('rt_path', ['transit', 'emission'])
def test_eval_bands_column_order(eng, rt_path):
    """TableSpectrum.eval_bands orders the columns by the first walker's optical depth
    (column_order='auto'; transit: rows of the matrix-core kernel, emission: layers of the fused
    kernel): band fluxes bit for bit those of the grid order (column_order=None) and of an order
    given by the caller; a non-permutation is refused."""
    import torch
    from pyratbay_amd import synth
    rng = np.random.default_rng(77)
    nspec, ntemp, L, W, nw = 3, 6, 33, 2049, 10
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    wn = g['wn']
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    bands = [(lo, np.ones(hi - lo), 1.0) for lo, hi in ((5, 900), (800, 2040))]
    pb = eng.PassBands(wn, bands)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    radius = radius0[None] * (1 + 0.01 * rng.uniform(-1, 1, (nw, 1)))
    args = [eng.dev(x) for x in (temps, dens)]
    out = {}
    for name, order in (('grid', None), ('auto', 'auto'), ('given', rng.permutation(W))):
        model = eng.TableSpectrum(etable, ttable, wn, radius0, 8.8e10, rt_path=rt_path,
                                  column_order=order)
        out[name] = model.eval_bands(*args, pb, radius=eng.dev(radius)).clone()
        assert (model.column_order is None) == (name == 'grid')
    assert torch.equal(out['auto'], out['grid']) and torch.equal(out['given'], out['grid'])
    assert bool(torch.isfinite(out['grid']).all())
    with pytest.raises(ValueError):
        eng.TableSpectrum(etable, ttable, wn, radius0, 8.8e10, column_order=np.zeros(W, int))


@pytest.mark.parametrize('L,itop', [(150, 0), (129, 0), (150, 30), (2, 1)])
def test_eval_bands_beyond_the_ordered_kernel(eng, orc, L, itop):
    """Shapes the depth-ordered transit kernel does not take (more than 128 impact parameters, or
    fewer than 2): the default column_order='auto' must fall through to the grid-order kernels
    (it used to raise on the first eval_bands call), an explicit order likewise; (150, 30) has 120
    impact parameters and does order.  Band fluxes against the oracle chain."""
    import torch
    from pyratbay_amd import synth
    rng = np.random.default_rng(300 + L + itop)
    nspec, ntemp, W, nw = 2, 5, 321, 5
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    wn = g['wn']
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    pb = eng.PassBands(wn, [(3, np.ones(300), 1.0)])
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    out = {}
    for name, order in (('grid', None), ('auto', 'auto'), ('given', rng.permutation(W))):
        model = eng.TableSpectrum(etable, ttable, wn, radius0, 8.8e10, itop=itop,
                                  column_order=order)
        out[name] = model.eval_bands(eng.dev(temps), eng.dev(dens), pb).clone()
        if name == 'auto':
            assert (model.column_order is not None) == (2 <= L - itop <= 128)
    assert torch.equal(out['auto'], out['grid']) and torch.equal(out['given'], out['grid'])
    for w in (0, nw - 1):
        ec = np.zeros((L, W))
        orc.interp_ec(ec, etable, ttable, temps[w], dens[w], 0, L)
        depth, ideep = orc.optical_depth_transit(ec, radius0, itop, L, 10.0)
        spec = orc.transmission(depth, radius0, 8.8e10, ideep, itop)
        want = np.trapezoid(spec[3:303], wn[3:303])
        np.testing.assert_allclose(host(out['grid'][w])[0], want, rtol=1e-11)


@pytest.mark.parametrize('L,itop,W', [(80, 0, 3000), (33, 2, 700), (128, 0, 515), (17, 0, 256)])
def test_tile_limited_batch(eng, orc, L, itop, W):
    """The layers nobody reads (pb_interp_ec_batch_limited + pb_transit_spectrum_limited /
    pb_emission_flux_limited + the device-gated repair).  Limits that are right, limits that are far too low for some blocks
    (every walker overruns: all flagged, all repaired) and limits too low for a few walkers only:
    the spectra are bit for bit those of the unlimited ordered path; the interpolation leaves the
    layers beyond a block's limit (and above itop) untouched; flags name exactly the walkers
    whose columns ran past a limit."""
    import torch
    rng = np.random.default_rng(1000 + L + W)
    nspec, ntemp, nw = 3, 6, 9
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-3, 3, (nspec, 1, 1, W))
    radius0 = np.linspace(8.0e9, 7.0e9, L)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    dens[3] *= 30.0                                   # one walker far more opaque ...
    dens[5] /= 30.0                                   # ... and one far more transparent than the base
    et, tt = eng.dev(etable), eng.dev(ttable)
    td, dd = eng.dev(temps), eng.dev(dens)
    rad = eng.dev(np.tile(radius0, (nw, 1)))
    path = eng.transit_path_device(rad, itop)
    # order by walker 0 and derive its exact per-column tiles
    ec0 = eng.interp_ec_batch(et, tt, td, dd)
    _, _, ideep = eng.transit_spectrum_batch(ec0, path, rad, 8.8e10, itop, L, 10.0, want_depth=True)
    order = torch.sort(ideep[0], stable=True).indices
    eto = et[..., order].contiguous()
    col = order.to(torch.int32)
    want = eng.transit_spectrum_ordered(eng.interp_ec_batch(eto, tt, td, dd), path, rad, col,
                                        8.8e10, itop, L, 10.0)
    nblk = -(-W // 256)
    ntiles = -(-(L - itop) // 16)
    need = torch.zeros(nblk * 256, dtype=torch.int64, device='cuda')
    allw = ideep[:, order].max(dim=0).values.to(torch.int64)           # deepest over the walkers
    need[:W] = allw
    need[W:] = allw[-1]
    exact = torch.clamp((need.view(nblk, 256).max(dim=1).values - itop) // 16, 0, ntiles - 1)
    base = torch.zeros(nblk * 256, dtype=torch.int64, device='cuda')
    base[:W] = ideep[0, order]
    base[W:] = base[W - 1]
    base_tile = torch.clamp((base.view(nblk, 256).max(dim=1).values - itop) // 16, 0, ntiles - 1)
    for name, tile in (('exact', exact), ('base model, no margin', base_tile),
                       ('far too low', torch.zeros_like(exact))):
        tile = tile.to(torch.int32).contiguous()
        flags = torch.zeros(nw + 1, dtype=torch.int32, device='cuda')
        ec = torch.full((nw, L, W), -7.0, dtype=torch.float64, device='cuda')
        iwork = torch.empty(nw * L * 17 + 8, dtype=torch.float64, device='cuda')
        eng.interp_ec_batch(eto, tt, td, dd, out=ec, tile_limit=tile, row0=itop, work=iwork)
        # what was written: exactly the layers itop ... itop + 16 (tile + 1) - 1 of each block
        # (a workgroup covers several blocks: up to the largest limit among them)
        full = eng.interp_ec_batch(eto, tt, td, dd)
        written = ec != -7.0
        lay = torch.arange(L, device='cuda')[None, :, None]
        blk = (torch.arange(W, device='cuda') // 256)[None, None, :]
        must = (lay >= itop) & (lay <= itop + 16 * (tile.to(torch.int64)[blk] + 1) - 1)
        assert bool((written | ~must).all()), name            # every wanted layer is there
        assert torch.equal(ec[written], full[written]), name  # ... with the right values
        assert not bool(written[:, :itop].any()), name
        if name == 'far too low' and ntiles > 1:
            assert not bool(written.all())
        twork = torch.empty(eng._capi.lib().pb_transit_work_doubles(L, itop, L, W, nw),
                            dtype=torch.float64, device='cuda')
        got = eng.transit_spectrum_ordered(ec, path, rad, col, 8.8e10, itop, L, 10.0,
                                           tile_limit=tile, flags=flags, work=twork)
        f = flags.cpu().numpy()
        assert f[nw] == int(f[:nw].any())
        # which walkers have a wavefront (32 ordered columns) still open beyond its block's tile
        wtile = ((ideep[:, order].to(torch.int64) - itop) // 16).cpu().numpy()
        lim_col = tile.cpu().numpy()[np.arange(W) // 256]
        c32 = (np.arange(W) // 32)
        over = np.zeros(nw, bool)
        for w in range(nw):
            wave_need = np.maximum.reduceat(wtile[w], np.flatnonzero(np.diff(c32, prepend=-1)))
            wave_lim = lim_col[np.flatnonzero(np.diff(c32, prepend=-1))]
            over[w] = np.any(wave_need > wave_lim)
        assert np.array_equal(f[:nw] != 0, over), (name, f, over)
        if name == 'exact':
            assert not f.any() and torch.equal(got, want)
        # the gated repair: full interpolation if any flag, transit of the flagged walkers
        eng.interp_ec_batch(eto, tt, td, dd, out=ec, gate=flags[nw:nw + 1], work=iwork)
        eng.transit_spectrum_ordered(ec, path, rad, col, 8.8e10, itop, L, 10.0, gate=flags,
                                     out=got, work=twork)
        assert torch.equal(got, want), name
        if f[nw]:
            assert torch.equal(ec, full)                      # (the repair rewrote every layer)
    # the same through TableSpectrum.eval_bands (walker 0 is the base model of the auto order)
    from pyratbay_amd import synth
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    pb = eng.PassBands(g['wn'], [(1, np.ones(W - 2), 1.0)])
    for rt_path in ('transit', 'emission'):
        res = {}
        for order_kw in (None, 'auto'):
            model = eng.TableSpectrum(etable, ttable, g['wn'], radius0, 8.8e10, itop=itop,
                                      rt_path=rt_path, column_order=order_kw)
            model.tile_margin = 0
            res[order_kw] = model.eval_bands(td, dd, pb).clone()
            if order_kw == 'auto' and W >= 64 and ntiles > 1 and rt_path == 'transit':
                assert model.tile_limit is not None
        assert torch.equal(res[None], res['auto']), rt_path
        assert bool(torch.isfinite(res[None]).all())


def test_fused_transit_equals_split(eng):
    """The fused column kernel against the two-kernel form it replaced (a separate process so
    that PB_TRANSIT=split is read afresh): same depth, ideep, spectrum bits."""
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, "tests")
import cases
from pyratbay_amd import engine as eng
c = cases.column_case(seed=5, nlayers=40, nwave=3000)
out = {}
for itop, ib in ((0, 40), (3, 31)):
    path = eng.dev(eng.pack_raypath(eng.transit_path(c["radius"], itop), itop))
    s, d, i = eng.transit_spectrum(eng.dev(c["ec"]), path, eng.dev(c["radius"]), c["rstar"], itop, ib, 10.0)
    out[f"s{itop}"], out[f"d{itop}"], out[f"i{itop}"] = s.cpu().numpy(), d.cpu().numpy(), i.cpu().numpy()
np.savez(sys.argv[1], **out)
'''
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for mode in ('fused', 'split'):
            env = dict(os.environ, PB_TRANSIT=mode)
            f = os.path.join(tmp, mode + '.npz')
            subprocess.run([sys.executable, '-c', code, f], check=True, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            res[mode] = dict(np.load(f))
        for k in res['fused']:
            assert np.array_equal(res['fused'][k], res['split'][k]), k


def test_eval_bands_64_walkers_vs_oracle(eng, orc):
    """A 64-walker batch with per-walker radius profiles: band fluxes against the oracle chain
    interp_ec -> transit_path -> optical depth -> transmission -> trapezoid, out-of-range
    temperatures rejected with +inf, and the batch equals the single-walker eval() bit for bit."""
    from pyratbay_amd import synth
    rng = np.random.default_rng(21)
    nspec, ntemp, L, W, nw = 4, 10, 20, 3001, 64
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    wn = g['wn']
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-1, 1, (nspec, 1, 1, W))
    base_radius = np.linspace(8.0e9, 7.0e9, L)
    rstar = 8.8e10
    model = eng.TableSpectrum(etable, ttable, wn, base_radius, rstar)
    bands = []
    for lo, hi in ((20, 900), (800, 2100), (2000, 2990)):
        resp = np.exp(-np.linspace(-1.5, 1.5, hi - lo)**2)
        bands.append((lo, resp, 1.0 / np.trapezoid(resp, wn[lo:hi])))
    pb = eng.PassBands(wn, bands)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-7, -3, (nw, 1, nspec))
    radius = np.array([hydro_radius(rng, L, 0.01) for _ in range(nw)])
    temps[5, 7] = 3000.5                                   # above the table
    temps[40, 0] = 299.0                                   # below the table
    temps[41, 3] = np.nan
    got = host(model.eval_bands(eng.dev(temps), eng.dev(dens), pb, radius=eng.dev(radius),
                                chunk=24))
    for w in (5, 40, 41):
        assert np.all(np.isinf(got[w])) and np.all(got[w] > 0)
    ok = [w for w in range(nw) if w not in (5, 40, 41)]
    assert np.all(np.isfinite(got[ok]))
    for w in ok[::5]:
        ec = np.zeros((L, W))
        orc.interp_ec(ec, etable, ttable, temps[w], dens[w], 0, L)
        depth, ideep = orc.optical_depth_transit(ec, radius[w], 0, L, 10.0)
        spec = orc.transmission(depth, radius[w], rstar, ideep, 0)
        want = [np.trapezoid(spec[s:s + len(r)] * r, wn[s:s + len(r)]) * h for s, r, h in bands]
        np.testing.assert_allclose(got[w], want, rtol=1e-11)
    # the batch and the one-walker path do the same arithmetic
    for w in (0, 33):
        model.set_radius(radius[w])
        spec = model.eval(temps[w], eng.dev(dens[w]))
        one = host(pb.partial_integrate(spec) * pb.heights)
        np.testing.assert_allclose(got[w], one, rtol=1e-13)
    # shared radius (the model's own): one ray-path table for the whole batch
    model.set_radius(base_radius)
    got2 = host(model.eval_bands(eng.dev(temps[:8]), eng.dev(dens[:8]), pb))
    spec = model.eval(temps[1], eng.dev(dens[1]))
    np.testing.assert_allclose(got2[1], host(pb.partial_integrate(spec) * pb.heights), rtol=1e-13)


def test_eval_bands_emission_batch(eng, orc):
    """The walker batch in emission geometry (plane-parallel optical depth + Planck + intensity +
    quadrature sum fused in one pass, pb_emission_flux_batch): equal to the one-walker eval()
    bit for bit and to the oracle chain at 1e-11; rejects as in the transit case."""
    from pyratbay_amd import synth
    rng = np.random.default_rng(5)
    nspec, ntemp, L, W, nw = 3, 8, 18, 2001, 21
    g = synth.spectral_grid(4000.0, 4000.0 + (W - 1) * 0.05 + 0.01, 0.05, 12)
    wn = g['wn']
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-27, -21, (nspec, ntemp, L, 1)) * \
        10.0**rng.uniform(-1, 1, (nspec, 1, 1, W))
    base_radius = np.linspace(8.0e9, 7.0e9, L)
    raygrid = np.radians([0.0, 20.0, 40.0, 60.0, 80.0])
    mu = np.cos(raygrid)
    bounds = np.linspace(0, 0.5 * np.pi, len(raygrid) + 1)
    bounds[1:-1] = 0.5 * (raygrid[:-1] + raygrid[1:])
    weights = np.pi * (np.sin(bounds[1:])**2 - np.sin(bounds[:-1])**2)
    model = eng.TableSpectrum(etable, ttable, wn, base_radius, 8.8e10, rt_path='emission',
                              quadrature_mu=mu, quadrature_weights=weights)
    bands = []
    for lo, hi in ((20, 900), (800, 1990)):
        resp = np.exp(-np.linspace(-1.5, 1.5, hi - lo)**2)
        bands.append((lo, resp, 1.0 / np.trapezoid(resp, wn[lo:hi])))
    pb = eng.PassBands(wn, bands)
    temps = 1500.0 * (1 + 0.1 * rng.uniform(-1, 1, (nw, 1))) * np.linspace(0.8, 1.2, L)
    dens = (press / temps)[:, :, None] * 7.2e21 * 10.0**rng.uniform(-6, -2, (nw, 1, nspec))
    radius = np.array([hydro_radius(rng, L, 0.01) for _ in range(nw)])
    temps[4, 2] = 3001.0
    got = host(model.eval_bands(eng.dev(temps), eng.dev(dens), pb, radius=eng.dev(radius),
                                chunk=8))
    assert np.all(np.isinf(got[4]))
    # the same model without an explicit quadrature: the reference's default raygrid (it used to
    # upload None and return NaN)
    plain = eng.TableSpectrum(etable, ttable, wn, base_radius, 8.8e10, rt_path='emission')
    got_plain = host(plain.eval_bands(eng.dev(temps), eng.dev(dens), pb, radius=eng.dev(radius),
                                      chunk=8))
    assert np.array_equal(got_plain, got)
    with pytest.raises(ValueError):
        eng.TableSpectrum(etable, ttable, wn, base_radius, 8.8e10, rt_path='emission',
                          quadrature_mu=mu)
    for w in (0, 7, 20):
        model.set_radius(radius[w])
        spec = model.eval(temps[w], eng.dev(dens[w]))
        one = host(pb.partial_integrate(spec) * pb.heights)
        np.testing.assert_allclose(got[w], one, rtol=1e-14)
        ec = np.zeros((L, W))
        orc.interp_ec(ec, etable, ttable, temps[w], dens[w], 0, L)
        depth = np.zeros((L, W))
        ideep = np.zeros(W, np.int32)
        orc.plane_parallel_optical_depth(depth, ideep, ec, -np.diff(radius[w]), 10.0, 0, L)
        B = orc.blackbody_wn_2D(wn, temps[w])
        inten = orc.intensity(depth, ideep, B, mu, 0)
        flux = np.sum(inten * weights[:, None], axis=0)
        want = [np.trapezoid(flux[s:s + len(r)] * r, wn[s:s + len(r)]) * h for s, r, h in bands]
        np.testing.assert_allclose(got[w], want, rtol=1e-11)


@pytest.mark.parametrize('nlayers,nwave,itop,ibottom', [(80, 1000, 0, None), (17, 70, 1, None),
                                                        (33, 257, 0, 30), (120, 300, 3, None),
                                                        (128, 130, 0, None), (2, 64, 0, None),
                                                        (16, 16, 0, None)])
def test_transit_on_the_matrix_cores(eng, orc, monkeypatch, nlayers, nwave, itop, ibottom):
    """k_transit_mfma (tau = Q . ec as 16x16x4 FP64 matrix products, the exit and the trapezoid
    on the accumulator layout) against the oracle's loops and against the vector kernel, over the
    shapes that select every instantiation: 1-8 row tiles, 4 or 2 column tiles per wavefront,
    rows and columns that are no multiples of 16, itop > 0, ibottom < L, columns that cross
    maxdepth in every row tile and columns that never do."""
    rng = np.random.default_rng(11)
    c = cases.column_case(seed=13, nlayers=nlayers, nwave=nwave)
    L, W, nw = c['nlayers'], c['nwave'], 3
    ibottom = L if ibottom is None else ibottom
    # spread the crossing layer over the whole column: scale every column by its own factor
    scale = 10.0**rng.uniform(-3, 3, W)
    ecs = np.array([c['ec'] * scale * 10.0**rng.uniform(-0.3, 0.3) for _ in range(nw)])
    radius = np.array([np.sort(c['radius'] * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    rad_d = eng.dev(radius)
    path = eng.transit_path_device(rad_d, itop)
    ec_d = eng.dev(ecs)
    got = host(eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, 10.0))
    monkeypatch.setenv('PB_TRANSIT_MFMA', '0')
    vec = host(eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], itop, ibottom, 10.0))
    np.testing.assert_allclose(got, vec, rtol=1e-13)
    stops = set()
    for w in range(nw):
        wd, wi = orc.optical_depth_transit(ecs[w], radius[w], itop, ibottom, 10.0)
        ws = orc.transmission(wd, radius[w], c['rstar'], wi, itop)
        np.testing.assert_allclose(got[w], ws, rtol=RTOL)
        stops |= set(np.unique(wi // 16))
    if L >= 80:
        assert len(stops) >= 2          # exits in several row tiles


@pytest.mark.gpu_experiments
@pytest.mark.parametrize('nlayers,nwave,itop,nmol', [(80, 1000, 0, 4), (80, 131, 2, 4),
                                                      (17, 99, 0, 1), (48, 40, 5, 3),
                                                      (33, 257, 0, 6), (120, 300, 3, 8),
                                                      (128, 130, 0, 4), (2, 64, 0, 2),
                                                      (16, 16, 0, 5), (96, 2, 1, 4)])
def test_table_transit_one_pass(eng, orc, monkeypatch, nlayers, nwave, itop, nmol):
    """pb_table_transit_batch -- interp_ec, optical depth and transmission of a batch in one pass,
    the interpolated extinction formed in registers as the operand of the matrix products -- against
    the two-pass form (pb_interp_ec_batch + pb_transit_spectrum_batch on a stored ec; 1e-13: the one
    pass forms the sums of interp_ec with fused multiply-adds) and against the oracle's interp_ec +
    optdepth loop + transmission (1e-12).  Shapes: 1-8 row tiles, 1-8 species (both register
    widths, species beyond nmol padded), ragged rows / columns, itop > 0, the smallest grid,
    temperatures on table nodes and at both table ends, per-walker radii."""
    rng = np.random.default_rng(23)
    L, W, nw, ntemp = nlayers, nwave, 5, 6
    assert eng.table_transit_supported(nmol, ntemp, L, itop, L, W)
    ttable = np.linspace(300.0, 3000.0, ntemp)
    press = np.logspace(-6, 2, L)
    # columns from transparent to opaque, so that the exits fall in every row tile
    colscale = 10.0**rng.uniform(-33.5, -27.0, W)
    etable = (10.0**rng.uniform(-0.3, 0.3, (nmol, ntemp, L, W)) * colscale *
              np.linspace(1.0, 2.0, ntemp)[None, :, None, None])
    etable[:, :, :, :2] = 0.0
    temps = rng.uniform(300.0, 3000.0, (nw, L))
    temps[0] = ttable[rng.integers(0, ntemp, L)]
    temps[1] = 300.0
    temps[2] = 3000.0
    dens = press[None, :, None]**0.9 * 10.0**rng.uniform(17, 18, (nw, L, nmol))
    radius = np.array([np.sort(np.linspace(8.0e9, 7.0e9, L) * (1 + 0.01 * rng.uniform(-1, 1)) +
                               rng.uniform(-1e5, 1e5, L))[::-1] for _ in range(nw)])
    rstar = 8.8e10
    et, tt = eng.dev(etable), eng.dev(ttable)
    td, dd, rd = eng.dev(temps), eng.dev(dens), eng.dev(radius)
    path = eng.transit_path_device(rd, itop)
    got = host(eng.table_transit_batch(et, tt, td, dd, path, rd, rstar, itop, L, 10.0))
    ec = eng.interp_ec_batch(et, tt, td, dd)
    two = host(eng.transit_spectrum_batch(ec, path, rd, rstar, itop, L, 10.0))
    np.testing.assert_allclose(got, two, rtol=1e-13)
    # (default: k_walker_order pairs neighbours in bracket order and picks the two-walker kernel
    # unless the pairs' brackets differ in more than a fifth of the K-steps -- these random
    # temperatures do; PB_TT_PAIR=2 forces it: every K-step then takes the exception path with the
    # second walker's own loads, an odd walker is left over; PB_TT_PAIR=0: one walker per wavefront)
    monkeypatch.setenv('PB_TT_PAIR', '2')
    forced = host(eng.table_transit_batch(et, tt, td, dd, path, rd, rstar, itop, L, 10.0))
    np.testing.assert_allclose(forced, two, rtol=1e-13)
    monkeypatch.setenv('PB_TT_PAIR', '0')
    single = host(eng.table_transit_batch(et, tt, td, dd, path, rd, rstar, itop, L, 10.0))
    monkeypatch.delenv('PB_TT_PAIR')
    np.testing.assert_allclose(single, two, rtol=1e-13)
    np.testing.assert_allclose(got, single, rtol=1e-13)
    # an even batch whose walkers share every bracket (one set of slice loads per pair)
    t2 = np.tile(temps[3:4], (4, 1)) * (1 + 1e-6 * np.arange(4)[:, None])
    d2 = np.tile(dens[3:4], (4, 1, 1)) * (1 + 0.1 * np.arange(4)[:, None, None])
    r2 = np.tile(radius[3:4], (4, 1))
    t2d, d2d, r2d = eng.dev(t2), eng.dev(d2), eng.dev(r2)
    p2 = eng.transit_path_device(r2d, itop)
    a = host(eng.table_transit_batch(et, tt, t2d, d2d, p2, r2d, rstar, itop, L, 10.0))
    b = host(eng.transit_spectrum_batch(eng.interp_ec_batch(et, tt, t2d, d2d), p2, r2d, rstar,
                                        itop, L, 10.0))
    np.testing.assert_allclose(a, b, rtol=1e-13)
    # ... and one whose pairs differ in a few layers only (the default picks the two-walker
    # kernel and takes the exception path in those K-steps)
    t3 = t2.copy()
    t3[1, L // 2] = temps[4, L // 2]
    t3[2, itop] = temps[4, itop]
    t3[3, L - 1] = temps[4, L - 1]
    t3d = eng.dev(t3)
    a = host(eng.table_transit_batch(et, tt, t3d, d2d, p2, r2d, rstar, itop, L, 10.0))
    b = host(eng.transit_spectrum_batch(eng.interp_ec_batch(et, tt, t3d, d2d), p2, r2d, rstar,
                                        itop, L, 10.0))
    np.testing.assert_allclose(a, b, rtol=1e-13)
    stops = set()
    for w in range(nw):
        want_ec = np.zeros((L, W))
        orc.interp_ec(want_ec, etable, ttable, temps[w], dens[w], 0, L)
        wd, wi = orc.optical_depth_transit(want_ec, radius[w], itop, L, 10.0)
        ws = orc.transmission(wd, radius[w], rstar, wi, itop)
        np.testing.assert_allclose(got[w], ws, rtol=RTOL)
        stops |= set(np.unique(wi // 16))
    if L >= 80 and W >= 100:
        assert len(stops) >= 2


@pytest.mark.gpu_experiments
def test_eval_bands_one_pass_equals_two_passes(eng):
    """TableSpectrum.eval_bands with one_pass = True (pb_table_transit_batch) against its default
    (interpolation and optical depth as two passes over a stored ec): band fluxes to 1e-13, the
    rejected walker +inf in both."""
    import torch
    rng = np.random.default_rng(5)
    nmol, ntemp, L, W, nw = 4, 6, 40, 3000, 9
    ttable = np.linspace(300.0, 3000.0, ntemp)
    wn = np.linspace(4000.0, 4150.0, W)
    press = np.logspace(-6, 2, L)
    etable = 10.0**rng.uniform(-30, -24, (nmol, ntemp, L, W))
    temps = rng.uniform(900.0, 1800.0, (nw, L))
    temps[3, 5] = 3200.0                                   # outside the table: rejected
    dens = press[None, :, None]**0.9 * 10.0**rng.uniform(14, 15, (nw, L, nmol))
    radius = np.array([np.sort(np.linspace(8.0e9, 7.0e9, L) * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                       for _ in range(nw)])
    model = eng.TableSpectrum(etable, ttable, wn, radius[0], 8.8e10, rt_path='transit')
    bands = eng.PassBands(wn, [(100, np.ones(500), 1.0), (1500, np.linspace(0.2, 1.0, 900), 0.5)])
    td, dd, rd = eng.dev(np.clip(temps, 300.0, 3000.0)), eng.dev(dens), eng.dev(radius)
    td[3, 5] = 3200.0
    two = host(model.eval_bands(td, dd, bands, radius=rd))
    model.one_pass = True
    one = host(model.eval_bands(td, dd, bands, radius=rd))
    assert np.all(np.isinf(two[3])) and np.all(np.isinf(one[3]))
    keep = np.arange(nw) != 3
    np.testing.assert_allclose(one[keep], two[keep], rtol=1e-13)


@pytest.mark.gpu_experiments
def test_table_transit_refuses_other_shapes(eng):
    """One impact parameter, more than 8 row tiles, a one-sample grid or a table whose species blocks
    exceed 32-bit byte offsets have no one-pass form:
    pb_table_transit_supported says so and the entry returns an error instead of a wrong launch."""
    assert not eng.table_transit_supported(4, 3, 1, 0, 1, 100)
    assert not eng.table_transit_supported(4, 3, 130, 0, 130, 100)
    assert not eng.table_transit_supported(4, 3, 80, 0, 80, 1)
    assert not eng.table_transit_supported(9, 3, 80, 0, 80, 100)
    assert not eng.table_transit_supported(4, 30, 80, 0, 80, 1000001)   # 19 GB per species
    assert eng.table_transit_supported(4, 3, 130, 2, 130, 100)
    import torch
    et = torch.zeros((4, 3, 130, 10), dtype=torch.float64, device='cuda')
    tt = eng.dev(np.array([300.0, 700.0, 1100.0]))
    temps = torch.full((2, 130), 500.0, dtype=torch.float64, device='cuda')
    dens = torch.ones((2, 130, 4), dtype=torch.float64, device='cuda')
    rad = eng.dev(np.tile(np.linspace(8e9, 7e9, 130), (2, 1)))
    path = eng.transit_path_device(rad, 0)
    with pytest.raises(Exception, match='one-pass'):
        eng.table_transit_batch(et, tt, temps, dens, path, rad, 8.8e10, 0, 130, 10.0)


@pytest.mark.gpu_experiments
@pytest.mark.parametrize('mode', ['0', '2'])
def test_table_transit_reads_inside_the_table(eng, monkeypatch, mode):
    """The one-pass kernels address the table with clamped rows / columns / species instead of
    predicates.  Here the table is a view into a buffer whose surroundings are NaN: a read outside
    the table would reach a sum (0 x NaN = NaN) and show in the spectrum.  Shapes whose last
    wavefront, last pair and last row tile are ragged; one walker per wavefront (mode 0) and two
    (mode 2: forced, so that the second walker's own loads run in every K-step)."""
    import torch
    rng = np.random.default_rng(31)
    monkeypatch.setenv('PB_TT_PAIR', mode)
    for nmol, L, W, nw, itop in ((4, 80, 1001, 5, 0), (3, 37, 131, 4, 3), (1, 18, 33, 3, 0),
                                 (4, 66, 2, 2, 1)):
        ntemp = 5
        ttable = np.linspace(300.0, 3000.0, ntemp)
        n = nmol * ntemp * L * W
        pad = 8192
        buf = torch.full((n + 2 * pad,), float('nan'), dtype=torch.float64, device='cuda')
        colscale = 10.0**rng.uniform(-33.5, -27.0, W)
        etable = 10.0**rng.uniform(-0.3, 0.3, (nmol, ntemp, L, W)) * colscale
        buf[pad:pad + n] = eng.dev(etable).view(-1)
        et = buf[pad:pad + n].view(nmol, ntemp, L, W)
        temps = rng.uniform(300.0, 3000.0, (nw, L))
        temps[0] = 3000.0
        press = np.logspace(-6, 2, L)
        dens = press[None, :, None]**0.9 * 10.0**rng.uniform(17, 18, (nw, L, nmol))
        radius = np.array([np.sort(np.linspace(8.0e9, 7.0e9, L) * (1 + 0.01 * rng.uniform(-1, 1)))[::-1]
                           for _ in range(nw)])
        tt, td, dd, rd = eng.dev(ttable), eng.dev(temps), eng.dev(dens), eng.dev(radius)
        path = eng.transit_path_device(rd, itop)
        got = host(eng.table_transit_batch(et, tt, td, dd, path, rd, 8.8e10, itop, L, 10.0))
        assert np.all(np.isfinite(got)), (nmol, L, W, nw, itop)
        two = host(eng.transit_spectrum_batch(eng.interp_ec_batch(et, tt, td, dd), path, rd,
                                              8.8e10, itop, L, 10.0))
        np.testing.assert_allclose(got, two, rtol=1e-13)
