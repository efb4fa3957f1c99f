"""compute_opacity (runmode=opacity as one batched GPU job) and the table-based eval path,
against the oracle and against the line-by-line path.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup():
    from pyratbay_amd import engine, synth
    engine.require_gpu()
    case = synth.lbl_case(1201, 6, 3000, wnosamp=24, nlor=16, ndop=8, extent=60.0,
                          cutoff=4.0, niso=2, seed=4)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = engine.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = engine.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                     vg['cutoff'], 1e-30, max_layers=32)
    return engine, synth, case, vt, ll, lbl


def test_compute_opacity_vs_oracle(setup, orc, tmp_path):
    engine, synth, case, vt, ll, lbl = setup
    from pyratbay_amd import opacity_table as ot
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    tgrid = np.array([800.0, 1300.0, 1800.0, 2300.0])
    pf = np.array([synth.partition_function(tgrid)] * 2)               # [niso, ntemp]
    etable = ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf,
                                chunk_bytes=5 * lbl.nwave * 8)           # forces chunking
    assert etable.shape == (1, 4, 6, g['nwave'])
    got = etable.cpu().numpy()[0]
    profile = vt.flat()
    for itemp in (0, 3):
        for ilayer in (0, 2, 5):
            t = tgrid[itemp]
            dens = atm['vmr'][ilayer] * atm['press'][ilayer] * synth.BAR / (synth.K_B * t)
            want = np.zeros((1, g['nwave']))
            orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                           g['wn'], g['own'], g['divisors'], dens, atm['mol_radius'],
                           atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                           pf[:, itemp].copy(), iso['isoiext'], ln['lwn'], ln['elow'], ln['gf'],
                           ln['lid'], vg['cutoff'], 1e-30, t, 0, 0, 0)
            np.testing.assert_allclose(got[itemp, ilayer], want[0], rtol=1e-10)
    path = str(tmp_path / 'table.npz')
    ot.write_opacity(path, 'H2O', tgrid, atm['press'], g['wn'], etable[0])
    units, species, t, p, w, o = ot.read_opacity(path)
    assert species == 'H2O' and o.shape == (4, 6, g['nwave'])
    np.testing.assert_array_equal(o, got)


def test_compute_opacity_on_a_constant_resolution_grid(orc):
    """The usual grid of an opacity table: constant resolving power (the reference's `resolution`
    mode, _extcoeff.c:320-326).  compute_opacity() with the plan's per-layer dynamic grids
    (gather mode 'dynamic': the cells of a chunk in runs of equal oversampling factor) against the
    direct gather (1e-12) and the oracle (1e-10), several cells, chunked."""
    from pyratbay_amd import engine, synth, opacity_table as ot
    case = synth.lbl_case(1501, 5, 3000, wnosamp=24, nlor=16, ndop=8, extent=60.0, cutoff=4.0,
                          niso=2, seed=6, resolution=45000.0)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24, 2)
    ll = engine.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = engine.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                     vg['cutoff'], 1e-30, resolution=True, max_layers=16)
    tgrid = np.array([700.0, 1200.0, 1900.0])
    pf = np.array([synth.partition_function(tgrid)] * 2)
    tables = {}
    for mode in ('auto', 'dynamic'):
        lbl.set_gather_mode(mode)
        tables[mode] = ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf,
                                          chunk_bytes=4 * lbl.nwave * 8).cpu().numpy()[0]
    assert lbl.last_gather_kernel == 'dynamic grids'
    assert np.array_equal(tables['auto'] == 0, tables['dynamic'] == 0)
    np.testing.assert_allclose(tables['dynamic'], tables['auto'], rtol=1e-12)
    profile = vt.flat()
    for itemp, ilayer in ((0, 0), (1, 4), (2, 2)):
        t = tgrid[itemp]
        dens = atm['vmr'][ilayer] * atm['press'][ilayer] * synth.BAR / (synth.K_B * t)
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], dens, atm['mol_radius'], atm['mol_mass'],
                       iso['isoimol'], iso['isomass'], iso['isoratio'], pf[:, itemp].copy(),
                       iso['isoiext'], ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'],
                       1e-30, t, 0, 0, 1)
        np.testing.assert_allclose(tables['dynamic'][itemp, ilayer], want[0], rtol=1e-10)


def test_table_path_equals_lbl_on_grid_nodes(setup):
    """With layer temperatures on table nodes the interpolated cross sections times the
    density reproduce the line-by-line extinction, and so does the spectrum."""
    engine, synth, case, vt, ll, lbl = setup
    from pyratbay_amd import opacity_table as ot
    g, atm, iso = case['grid'], case['atm'], case['iso']
    tgrid = np.array([900.0, 1200.0, 1500.0])
    temps = tgrid[[0, 1, 1, 2, 2, 0]]
    pf = np.array([synth.partition_function(tgrid)] * 2)
    etable = ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf)
    dens = atm['vmr'] * (atm['press'] / temps)[:, None] * synth.BAR / synth.K_B
    z = np.array([synth.partition_function(temps)] * 2)
    ec_lbl = lbl.extinction(engine.dev(temps), engine.dev(dens), engine.dev(z), add=True)
    model = engine.TableSpectrum(etable, tgrid, g['wn'], atm['radius'], atm['rstar'])
    spec = model.eval(temps, dens[:, 2:3])                   # H2O column density
    np.testing.assert_allclose(model.ec.cpu().numpy(), ec_lbl.cpu().numpy()[:, 0],
                               rtol=1e-12)
    depth, ideep = engine.optical_depth_transit(
        ec_lbl.view(6, -1), model.raypath, 0, 6, 10.0)
    want = engine.transmission(depth, ideep, model.radius, 0, atm['rstar'])
    np.testing.assert_allclose(spec.cpu().numpy(), want.cpu().numpy(), rtol=1e-12)


def test_table_eval_vs_oracle_between_nodes(setup, orc):
    engine, synth, case, vt, ll, lbl = setup
    g, atm = case['grid'], case['atm']
    rng = np.random.default_rng(8)
    nspec, ntemp, L, W = 3, 6, 6, g['nwave']
    ttable = np.linspace(500, 3000, ntemp)
    etable = 10.0**rng.uniform(-27, -20, (nspec, ntemp, L, W))
    temps = rng.uniform(600, 2900, L)
    dens = 10.0**rng.uniform(8, 16, (L, nspec))
    model = engine.TableSpectrum(etable, ttable, g['wn'], atm['radius'], atm['rstar'])
    spec = model.eval(temps, dens).cpu().numpy()
    ec = np.zeros((L, W))
    orc.interp_ec(ec, etable, ttable, temps, dens, 0, L)
    depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, L, 10.0)
    want = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    np.testing.assert_allclose(spec, want, rtol=1e-12)


def test_batched_walkers_bandflux(setup, orc):
    """C5 shape in miniature: several (T, abundance) vectors -> band fluxes; one walker
    outside the table's temperature range is rejected with inf."""
    engine, synth, case, vt, ll, lbl = setup
    g, atm = case['grid'], case['atm']
    rng = np.random.default_rng(12)
    nspec, ntemp, L, W = 2, 5, 6, g['nwave']
    ttable = np.linspace(600, 2600, ntemp)
    etable = 10.0**rng.uniform(-26, -21, (nspec, ntemp, L, W))
    model = engine.TableSpectrum(etable, ttable, g['wn'], atm['radius'], atm['rstar'])
    bands = []
    for lo, hi in ((20, 500), (450, 1100)):
        resp = np.exp(-np.linspace(-1, 1, hi - lo)**2)
        bands.append((lo, resp, 1.0 / np.trapezoid(resp, g['wn'][lo:hi])))
    pb = engine.PassBands(g['wn'], bands)
    nw = 4
    temps = rng.uniform(700, 2500, (nw, L))
    temps[2, 3] = 2700.0                                   # out of range -> rejected
    dens = 10.0**rng.uniform(9, 15, (nw, L, nspec))
    got = model.eval_bands(engine.dev(temps), engine.dev(dens), pb).cpu().numpy()
    assert np.all(np.isinf(got[2]))
    for w in (0, 1, 3):
        ec = np.zeros((L, W))
        orc.interp_ec(ec, etable, ttable, temps[w], dens[w], 0, L)
        depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, L, 10.0)
        spec = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
        want = [np.trapezoid(spec[s:s + len(r)] * r, g['wn'][s:s + len(r)]) * h
                for s, r, h in bands]
        np.testing.assert_allclose(got[w], want, rtol=1e-12)
