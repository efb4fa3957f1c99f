"""HIP line-by-line extinction (all layers per call, through the C ABI) against the
golden vectors of the compiled reference and against the oracle.  Needs an MI355X.

Tolerance: rtol 1e-10 on every non-zero sample and an identical zero pattern.  The
kernel sums the same terms in the same order with fma (one rounding instead of two)
and device exp(); the table itself agrees to ~1e-13 (test_gpu_voigt.py)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


def host(t):
    return t.cpu().numpy()


def build(eng, c, resolution, table_from_oracle, orc, ethresh=1e-30, cutoff=None):
    ownstep = c['own'][1] - c['own'][0]
    osamp = 12
    if table_from_oracle:
        size = c['size'].copy()
        index = np.zeros_like(size)
        profile = np.zeros(np.sum(2 * size + 1))
        orc.voigt_grid(profile, size, index, c['lorentz'], c['doppler'], ownstep)
        vt = eng.VoigtTable.from_flat(profile, size, index, c['lorentz'], c['doppler'], osamp)
    else:
        vt = eng.VoigtTable.build(c['lorentz'], c['doppler'], c['size'], ownstep, osamp)
    ll = eng.LineList(c['lwn'], c['elow'], c['gf'], c['lid'], 3, c['own'])
    atm, iso = c['atm'], c['iso']
    lbl = eng.LBL(vt, ll, c['wn'], c['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  c['cutoff'] if cutoff is None else cutoff, ethresh,
                  resolution=resolution, max_layers=16)
    return vt, ll, lbl


@pytest.mark.parametrize('mode,gather', [('step', 'global'), ('step', 'staged'),
                                         ('step', 'resident'), cases.exp('step', 'scatter'),
                                         cases.exp('step', 'rounds'), ('step', 'auto'),
                                         cases.exp('step', 'wave'),
                                         ('res', 'auto'), ('res', 'dynamic')])
@pytest.mark.parametrize('own_table', [False, True])
def test_g2_extinction_golden(eng, golden, orc, mode, own_table, gather):
    g = golden('g2_extinction')
    c = cases.extinction_inputs(resolution=(mode == 'res'))
    atm, iso = c['atm'], c['iso']
    nl = atm['nlayers']
    temp = eng.dev(atm['temp'])
    dens = eng.dev(atm['dens'])
    isoz = eng.dev(np.array([cases.iso_z(t, 3) for t in atm['temp']]).T)   # [niso, L]
    worst = 0.0
    plans = {}
    for k, (layer, add, cut, eth, skip) in enumerate(cases.extinction_variants()):
        key = (cut,)
        if key not in plans:
            plans[key] = build(eng, c, mode == 'res', not own_table, orc,
                               cutoff=c['cutoff'] if cut else 0.0)
        vt, ll, lbl = plans[key]
        lbl.set_gather_mode(gather)
        isoiext = iso['isoiext'].copy()
        if skip:
            isoiext[1] = -1
        lbl.set_isoiext(isoiext)
        lbl.set_ethresh(eth)
        ext = host(lbl.extinction(temp, dens, isoz, add=bool(add)))     # [L, rows, W]
        got = ext[layer]
        want = g[f'ext_{mode}'][k][:got.shape[0]]
        assert np.array_equal(got == 0, want == 0), f'variant {k}: zero pattern differs'
        nz = want != 0
        if nz.any():
            worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=f'variant {k}')
    print(f'{mode}/{gather} own_table={own_table}: max rel err vs reference = {worst:.2e}')
    assert ll.nadd > 0
    want_kernel = {'global': ('k_ext_resample',), 'staged': ('k_ext_staged',),
                   'resident': ('k_ext_resident+k_ext_resample',),
                   'scatter': ('k_ext_scatter',), 'rounds': ('k_ext_rounds',),
                   'dynamic': ('dynamic grids',), 'wave': ('k_ext_wave+k_ext_staged',),
                   'auto': ('k_ext_linterp',) if mode == 'res' else
                           ('k_ext_resident+k_ext_resample', 'k_ext_resident+k_ext_staged')}[gather]
    assert lbl.last_gather_kernel in want_kernel


def test_groups_match_oracle_counters(eng, orc):
    """co-add grouping on the host (pb_lines_create) == the reference's sequential pass."""
    c = cases.extinction_inputs()
    vt, ll, lbl = build(eng, c, False, True, orc)
    atm, iso = c['atm'], c['iso']
    size = vt.size.copy()
    index = vt.index.copy()
    profile = vt.flat()
    ext = np.zeros((2, len(c['wn'])))
    st = orc.extinction(ext, profile, size, index, c['lorentz'], c['doppler'], c['wn'],
                        c['own'], c['divisors'], atm['dens'][3], atm['mol_radius'],
                        atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                        cases.iso_z(atm['temp'][3], 3), iso['isoiext'], c['lwn'], c['elow'],
                        c['gf'], c['lid'], c['cutoff'], 1e-30, atm['temp'][3], 0, 0, 0,
                        return_stats=True)
    assert ll.nadd == st['nadd']
    assert ll.ngroups == st['neval'] + st['nskip']
    # dynamic-sampling factor and kmax per layer
    temp, dens = eng.dev(atm['temp']), eng.dev(atm['dens'])
    isoz = eng.dev(np.array([cases.iso_z(t, 3) for t in atm['temp']]).T)
    lbl.extinction(temp, dens, isoz, add=False)
    ofactor, kmax = lbl.last_state(atm['nlayers'], 2)
    for layer in range(atm['nlayers']):
        st = orc.extinction(ext, profile, size, index, c['lorentz'], c['doppler'], c['wn'],
                            c['own'], c['divisors'], atm['dens'][layer], atm['mol_radius'],
                            atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                            cases.iso_z(atm['temp'][layer], 3), iso['isoiext'], c['lwn'],
                            c['elow'], c['gf'], c['lid'], c['cutoff'], 1e-30,
                            atm['temp'][layer], 0, 0, 0, return_stats=True)
        assert ofactor[layer] == st['ofactor'], layer
    assert np.all(kmax > 0)


@pytest.mark.parametrize('gather', cases.gathers('global', 'staged', 'resident', 'scatter',
                                                  'rounds', 'wave'))
@pytest.mark.parametrize('nwave,nlines,niso', [(2, 1, 1), (65, 40, 1), (1025, 3000, 2),
                                               (4097, 20000, 4), (9001, 60000, 2)])
def test_synthetic_cases_vs_oracle(eng, orc, nwave, nlines, niso, gather):
    """Ragged grid sizes (tile edges), several isotopes, every layer; own table."""
    from pyratbay_amd import synth
    case = synth.lbl_case(nwave, 6, nlines, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=niso, seed=nwave)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=6)
    lbl.set_gather_mode(gather)
    ext = host(lbl.extinction(eng.dev(atm['temp']), eng.dev(atm['dens']),
                              eng.dev(iso['isoz']), add=True))
    profile = vt.flat()
    worst = 0.0
    for layer in range(6):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], g['own'], g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), iso['isoiext'],
                       ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], 1e-30,
                       atm['temp'][layer], 0, 1, 0)
        got = ext[layer]
        assert np.array_equal(got == 0, want == 0), layer
        nz = want != 0
        if nz.any():
            worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL)
    print(f'W={nwave} N={nlines} {gather}: max rel err vs oracle (same table) = {worst:.2e}')


@pytest.mark.parametrize('gather', cases.gathers('global', 'staged', 'resident', 'scatter',
                                                  'rounds', 'wave'))
def test_wavenumber_shards_concatenate(eng, orc, gather):
    """Shards [wbegin, wbegin+wcount) of the global grid reproduce the full spectrum
    bit for bit (no exchange between shards)."""
    from pyratbay_amd import synth
    case = synth.lbl_case(3001, 5, 6000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=2, seed=5)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=5)
    lbl.set_gather_mode(gather)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    full = host(lbl.extinction(t, d, z))
    bounds = [0, 377, 1024, 1025, 2500, 3001]
    parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
             for a, b in zip(bounds[:-1], bounds[1:])]
    assert np.array_equal(np.concatenate(parts, axis=2), full)
    # and twice the same call is bitwise reproducible
    assert np.array_equal(host(lbl.extinction(t, d, z)), full)


@pytest.mark.parametrize('gather', cases.gathers('auto', 'global', 'staged', 'resident',
                                                  'scatter', 'rounds', 'wave'))
@pytest.mark.parametrize('wnosamp,cutoff', [(24, 3.0), (12, 30.0), (6, 0.5)])
def test_shards_never_read_unwritten_records(eng, monkeypatch, gather, wnosamp, cutoff):
    """A wavenumber shard computes the records of the groups within reach of it only.  Every shard
    here runs on a FRESH plan whose record buffers start out filled with live records of strength
    1e300 (PB_POISON_RECORDS): a gather kernel that examined a group outside the window k_records
    wrote would turn the result into ~1e300 (with the round-2 window, two output samples too
    narrow for the kernels that find their candidates through the per-sample index, the resident
    kernel did -- whatever the allocation held, which is how a long-lived process faulted).  Also
    the two-phase form, whose other groups are not even visited."""
    from pyratbay_amd import synth
    case = synth.lbl_case(2001, 4, 5000, wnstep=0.2, wnosamp=wnosamp, nlor=12, ndop=6,
                          extent=150.0, cutoff=cutoff, niso=2, seed=31)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], wnosamp)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def plan():
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], 1e-30, max_layers=4)
        lbl.set_gather_mode(gather)
        return lbl

    monkeypatch.setenv('PB_STAGE_SPLIT', '1')            # one association of the sums
    full = host(plan().extinction(t, d, z))
    assert np.isfinite(full).all() and full.max() < 1e-3
    monkeypatch.setenv('PB_POISON_RECORDS', '1')
    bounds = [0, 433, 865, 866, 1500, 2001]
    for a, b in zip(bounds[:-1], bounds[1:]):
        part = host(plan().extinction(t, d, z, wbegin=a, wcount=b - a))
        assert np.array_equal(part, full[:, :, a:b]), (a, b)
        lbl = plan()
        out = eng.dev(np.zeros((4, 1, b - a)))
        lbl.extinction_begin(t, d, z, add=True, out=out, wbegin=a, wcount=b - a)
        lbl.extinction_end()
        # (the shard's own maxima: with ethresh = 1e-30 nothing is dropped either way)
        assert np.array_equal(host(out), full[:, :, a:b]), ('two-phase', a, b)


def test_empty_and_out_of_range_lines(eng):
    from pyratbay_amd import synth
    case = synth.lbl_case(513, 3, 50, wnosamp=12, nlor=8, ndop=4, extent=30.0, cutoff=2.0)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 12)
    for lwn in (ln['lwn'] + 1e4, ln['lwn'][:0]):          # all out of range / no lines
        n = len(lwn)
        ll = eng.LineList(lwn, ln['elow'][:n], ln['gf'][:n], ln['lid'][:n], 1, g['own'])
        assert ll.ngroups == 0
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], 1e-30, max_layers=3)
        ext = host(lbl.extinction(eng.dev(atm['temp']), eng.dev(atm['dens']),
                                  eng.dev(iso['isoz'])))
        assert ext.shape == (3, 1, 513) and np.all(ext == 0)


@pytest.mark.parametrize('gather', ['staged', 'global'])
def test_long_rows_vs_oracle(eng, orc, gather):
    """Phase rows longer than one staged LDS row (1024 samples): the staged kernel cuts them
    into chunks, (phase, chunk) pairs acting as phases.  Wide cutoff on a fine grid gives
    rows of ~3 200 samples here (4 chunks, the last one partial)."""
    from pyratbay_amd import synth
    case = synth.lbl_case(4097, 5, 2500, wnosamp=12, nlor=10, ndop=5, extent=4000.0,
                          cutoff=80.0, niso=2, seed=21)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 12)
    rowmax = int(np.max(2 * np.asarray(vt.size) + 1)) // 12
    assert rowmax > 2048, rowmax
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=5)
    lbl.set_gather_mode(gather)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    ext = host(lbl.extinction(t, d, z, add=True))
    assert lbl.last_gather_kernel == {'staged': 'k_ext_staged', 'global': 'k_ext_resample'}[gather]
    profile = vt.flat()
    worst = 0.0
    for layer in range(5):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], g['own'], g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), iso['isoiext'],
                       ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], 1e-30,
                       atm['temp'][layer], 0, 1, 0)
        got = ext[layer]
        assert np.array_equal(got == 0, want == 0), layer
        nz = want != 0
        worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL)
    print(f'long rows ({rowmax} samples) {gather}: max rel err vs oracle = {worst:.2e}')
    # shards of the chunked path concatenate exactly, too
    if gather == 'staged':
        parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
                 for a, b in ((0, 1500), (1500, 4097))]
        assert np.array_equal(np.concatenate(parts, axis=2), ext)


def test_staged_variants_agree(eng, monkeypatch):
    """The staged kernel's row staging (LDS-DMA, or through registers with PB_STAGE_DMA=0) and
    its tiling (PB_STAGE_S, PB_STAGE_SPLIT) are performance choices: the two staging forms
    give the same bits, and splitting the phases of a tile between workgroups changes only the
    association of the per-sample sums."""
    from pyratbay_amd import synth
    case = synth.lbl_case(9001, 6, 60000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=2, seed=77)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=6)
    lbl.set_gather_mode('staged')
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        out = host(lbl.extinction(t, d, z, add=True))
        assert lbl.last_gather_kernel == 'k_ext_staged'
        for k in env:
            monkeypatch.delenv(k)
        return out

    base = run(PB_STAGE_S=2, PB_STAGE_SPLIT=1)
    assert np.array_equal(run(PB_STAGE_S=2, PB_STAGE_SPLIT=1, PB_STAGE_DMA=0), base)
    assert np.array_equal(run(PB_STAGE_S=1, PB_STAGE_SPLIT=1), base)
    for split in (2, 3, 4, 8):
        for dma in (0, 1):
            got = run(PB_STAGE_S=2, PB_STAGE_SPLIT=split, PB_STAGE_DMA=dma)
            assert np.array_equal(got == 0, base == 0)
            np.testing.assert_allclose(got, base, rtol=1e-13)


def test_work_counters_of_last_launch(eng):
    """The device-side counters behind bench.py's roofline block (pb_lbl_last_work,
    pb_lbl_last_table_samples): multiplied profile samples partition exactly over wavenumber
    shards, the staged kernels issue whole 256-sample spans, and the distinct table samples of a
    call lie between the larger shard's and the sum of both."""
    from pyratbay_amd import synth
    case = synth.lbl_case(9001, 6, 60000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=2, seed=78)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=6)
    lbl.set_gather_mode('staged')
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    nwave = len(g['wn'])

    def counters(a, b):
        ext = host(lbl.extinction(t, d, z, add=True, wbegin=a, wcount=b - a))
        w = lbl.last_work()
        return w, lbl.last_table_samples(), ext

    whole, tab, ext = counters(0, nwave)
    assert 0 < whole['fma_lanes_useful'] <= whole['fma_lanes_issued']
    assert whole['fma_lanes_issued'] % 256 == 0
    assert 0 < whole["live_records"] <= 6 * ll.ngroups
    # every distinct table sample is multiplied at least once; none beyond the table per layer
    assert 0 < tab <= whole['fma_lanes_useful']
    assert tab <= 6 * vt.device_bytes // 8
    cut = 4097
    left, tab_l, _ = counters(0, cut)
    right, tab_r, _ = counters(cut, nwave)
    assert left['fma_lanes_useful'] + right['fma_lanes_useful'] == whole['fma_lanes_useful']
    assert max(tab_l, tab_r) <= tab <= tab_l + tab_r
    # a threshold that drops lines lowers every count
    lbl2 = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                   iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                   vg['cutoff'], 1e-3, max_layers=6)
    lbl2.set_gather_mode('staged')
    lbl2.extinction(t, d, z, add=True)
    w2, tab2 = lbl2.last_work(), lbl2.last_table_samples()
    assert w2['live_records'] < whole['live_records']
    assert w2['fma_lanes_useful'] < whole['fma_lanes_useful'] and tab2 <= tab


@pytest.mark.parametrize("seed", range(10))
def test_random_configurations(eng, orc, seed):
    """Randomly drawn grids, oversampling factors, profile extents, cutoffs, line densities
    and shard windows: the staged kernel (LDS-DMA rows; odd and even window starts, ragged
    last tiles, rows from a few to a few thousand samples, chunked rows) and the global gather
    agree with each other and, on three layers, with the oracle."""
    from pyratbay_amd import synth
    rng = np.random.default_rng(1000 + seed)
    nwave = int(rng.integers(3, 6000))
    nlayers = int(rng.integers(1, 7))
    nlines = int(rng.integers(1, 40000))
    niso = int(rng.integers(1, 4))
    wnosamp = int(rng.choice([6, 12, 24, 60]))
    extent = float(rng.choice([8.0, 40.0, 150.0, 600.0]))
    cutoff = float(rng.choice([0.0, 0.5, 3.0, 30.0]))
    wnstep = float(rng.choice([0.01, 0.05, 0.2]))
    case = synth.lbl_case(nwave, nlayers, nlines, wnstep=wnstep, wnosamp=wnosamp, nlor=12,
                          ndop=6, extent=extent, cutoff=cutoff, niso=niso, seed=seed)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], wnosamp)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=nlayers)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    a = int(rng.integers(0, g['nwave']))
    b = int(rng.integers(a, g['nwave'])) + 1
    out = {}
    for mode in cases.live('global', 'staged', 'rounds'):
        lbl.set_gather_mode(mode)
        out[mode] = host(lbl.extinction(t, d, z, add=True))
        out[mode + '_shard'] = host(lbl.extinction(t, d, z, add=True, wbegin=a, wcount=b - a))
        assert np.array_equal(out[mode + '_shard'], out[mode][:, :, a:b]), mode
    assert np.array_equal(out['staged'] == 0, out['global'] == 0)
    np.testing.assert_allclose(out['staged'], out['global'], rtol=1e-12)
    if 'rounds' in out:
        assert np.array_equal(out['rounds'] == 0, out['global'] == 0)
        np.testing.assert_allclose(out['rounds'], out['global'], rtol=1e-12)
    profile = vt.flat()
    for layer in sorted(set([0, nlayers // 2, nlayers - 1])):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], g['own'], g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), iso['isoiext'],
                       ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], 1e-30,
                       atm['temp'][layer], 0, 1, 0)
        got = out['staged'][layer]
        assert np.array_equal(got == 0, want == 0), layer
        np.testing.assert_allclose(got, want, rtol=RTOL)


def reference_groups(lwn, lid, own):
    """The sequential pass of _extcoeff.c:230-262 restated for the test: leaders, members."""
    lo, hi, step = own[0], own[-1], own[1] - own[0]
    first, count, iown = [], [], []
    n = len(lwn)
    ln = 0
    while ln < n:
        v = lwn[ln]
        if v < lo or v > hi:
            ln += 1
            continue
        i = int((v - lo) / step)
        if i + 1 < len(own) and abs(v - own[i + 1]) < abs(v - own[i]):
            i += 1
        f, c = ln, 1
        while ln + 1 != n and lid[ln + 1] == lid[f] and lwn[ln + 1] <= hi \
                and abs(lwn[ln + 1] - own[i]) < step:
            ln += 1
            c += 1
        first.append(f)
        count.append(c)
        iown.append(i)
        ln += 1
    return np.array(first, np.int32), np.array(count, np.int32), np.array(iown, np.int32)


@pytest.mark.parametrize('kind', ['dense', 'ragged', 'one_iso_empty', 'nothing', 'golden'])
def test_device_grouping_equals_reference_pass(eng, monkeypatch, kind):
    """pb_lines.hip (forward pointers + pointer doubling) against the reference's sequential
    co-adding pass and against the host loop of pb_lines_create: same leaders, member counts,
    fine indices, per-isotope segments and counters."""
    rng = np.random.default_rng(77)
    own = 5000.0 + np.arange(200001) * 0.0005            # 100 cm-1
    if kind == 'golden':
        c = cases.extinction_inputs()
        lwn, lid, own, niso = c['lwn'], c['lid'], c['own'], 3
    else:
        niso = 4
        parts, ids = [], []
        for i in range(niso):
            n = {'dense': 40000, 'ragged': 3000, 'one_iso_empty': 0 if i == 2 else 5000,
                 'nothing': 50}[kind]
            if kind == 'nothing':
                v = rng.uniform(6000, 6100, n)
            elif kind == 'dense':
                v = rng.uniform(4999.0, 5101.0, n)         # ~0.2 lines per fine step: chains
                v[:n // 4] = v[n // 4:2 * (n // 4)] + rng.uniform(-0.9, 0.9, n // 4) * 0.0005
            else:
                v = rng.uniform(4990.0, 5110.0, n)         # plenty outside on both sides
            parts.append(np.sort(v))
            ids.append(np.full(n, i, np.int32))
        lwn, lid = np.concatenate(parts), np.concatenate(ids)
    n = len(lwn)
    elow, gf = np.ones(n), np.ones(n)
    dev_ll = eng.LineList(lwn, elow, gf, lid, niso, own)
    assert dev_ll.grouped_on_device
    monkeypatch.setenv('PB_LINES_HOST', '1')
    host_ll = eng.LineList(lwn, elow, gf, lid, niso, own)
    monkeypatch.delenv('PB_LINES_HOST')
    assert not host_ll.grouped_on_device
    want = reference_groups(lwn, lid, own)
    for ll in (dev_ll, host_ll):
        first, count, iown, start = ll.groups()
        assert np.array_equal(first, want[0]) and np.array_equal(count, want[1])
        assert np.array_equal(iown, want[2])
        assert ll.ngroups == len(want[0]) and ll.nadd == int(np.sum(want[1] - 1))
        assert ll.ninrange == int(np.sum(want[1]))
        assert np.array_equal(start, np.searchsorted(lid[want[0]], np.arange(niso + 1)))
    if kind == 'dense':
        assert dev_ll.nadd > 10000 and np.max(want[1]) >= 4


def test_unsorted_lines_fall_back_to_host_grouping(eng, orc):
    """Isotopes may interleave in the list as long as each isotope's own lines ascend (the host
    loop groups them; the result equals the oracle's sequential pass).  A list that steps BACK
    within an isotope is refused: the reference's Doppler-index search is one-way
    (_extcoeff.c:278, utils.h:45-72), its result on such a list depends on the order, and the
    kernels evaluate the nearest index statelessly."""
    from pyratbay_amd import _capi, synth
    case = synth.lbl_case(3001, 4, 3000, wnosamp=24, nlor=14, ndop=7, extent=60.0, cutoff=3.0,
                          niso=2, seed=13)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    # interleave the two isotopes: ascending wavenumber overall
    o = np.argsort(ln['lwn'], kind='stable')
    mixed = {k: np.ascontiguousarray(ln[k][o]) for k in ('lwn', 'elow', 'gf', 'lid')}
    assert np.any(np.diff(mixed['lid']) < 0)
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'])
    ll = eng.LineList(mixed['lwn'], mixed['elow'], mixed['gf'], mixed['lid'], 2, g['own'])
    assert not ll.grouped_on_device
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'],
                  case['ethresh'], max_layers=4)
    ext = host(lbl.extinction(eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])))
    profile = vt.flat()
    for layer in range(4):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], mixed['lwn'], mixed['elow'],
                       mixed['gf'], mixed['lid'], vg['cutoff'], case['ethresh'],
                       atm['temp'][layer], 0, 1, 0)
        assert np.array_equal(ext[layer] == 0, want == 0)
        np.testing.assert_allclose(ext[layer], want, rtol=RTOL)
    rng = np.random.default_rng(3)
    own = 5000.0 + np.arange(20001) * 0.005
    lwn = rng.uniform(5000, 5100, 500)                    # one isotope, not sorted
    with pytest.raises(_capi.PbError, match='ascending wavenumber order'):
        eng.LineList(lwn, np.ones(500), np.ones(500), np.zeros(500, np.int32), 1, own)
    # (lines outside the fine grid are never evaluated: their order does not matter)
    lwn = np.concatenate([[6000.0, 4000.0], np.sort(lwn)])
    ll = eng.LineList(lwn, np.ones(502), np.ones(502), np.zeros(502, np.int32), 1, own)
    assert ll.ninrange == 500


@pytest.mark.parametrize('gather', cases.gathers('staged', 'global', 'rounds'))
@pytest.mark.parametrize('ethresh', [1e-30, 1e-3])
def test_two_phase_shards_with_kmax_exchange(eng, monkeypatch, gather, ethresh):
    """The multi-GPU form of a wavenumber shard (pb_lbl_extinction_begin / kmax all-reduce /
    _end): every "rank" derives records and strengths only for the groups within reach of its
    shard, the per-row maxima are combined with an element-wise integer MAX (what the RCCL
    all-reduce does), and the shards concatenate to the full-grid result bit for bit -- also
    with a threshold high enough to drop most lines, i.e. the exchanged maxima really are the
    global ones."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.setenv('PB_STAGE_SPLIT', '1')
    case = synth.lbl_case(9001, 5, 40000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=3, seed=8)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    iso['isoiext'] = np.array([0, 1, 0], np.int32)
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 3, g['own'])

    def plan():
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], ethresh, max_layers=5)
        lbl.set_gather_mode(gather)
        return lbl
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    for add in (True, False):
        full_plan = plan()
        full = host(full_plan.extinction(t, d, z, add=add))
        _, kmax_full = full_plan.last_state(5, 1 if add else 2)
        bounds = [0, 1300, 4096, 4097, 7000, 9001]
        ranks = [plan() for _ in bounds[:-1]]
        outs = [r.extinction_begin(t, d, z, add=add, wbegin=a, wcount=b - a)
                for r, a, b in zip(ranks, bounds[:-1], bounds[1:])]
        local = torch.stack([r.kmax_tensor().clone() for r in ranks])
        glob = local.max(dim=0).values                          # the all-reduce (MAX)
        nrow = 1 if add else 2
        assert np.array_equal(glob[:5 * nrow].view(torch.float64).cpu().numpy().reshape(5, nrow),
                              kmax_full)
        # most ranks do not see the strongest lines themselves: the exchange matters
        assert sum(not torch.equal(l_[:5 * nrow], glob[:5 * nrow]) for l_ in local) >= 3
        for r in ranks:
            r.kmax_tensor().copy_(glob)
            r.extinction_end()
        got = np.concatenate([host(o) for o in outs], axis=2)
        assert np.array_equal(got, full), (gather, add)
        if ethresh > 1e-10:
            # the threshold does drop lines here: the result differs from the unthresholded one
            loose = plan()
            loose.set_ethresh(1e-30)
            assert not np.array_equal(host(loose.extinction(t, d, z, add=add)), full)


@pytest.mark.parametrize('gather', cases.gathers('staged', 'rounds', 'global'))
@pytest.mark.parametrize('short_own', [False, True])
@pytest.mark.parametrize('long_rows', [False, True])
def test_windows_that_leave_the_grid(eng, orc, gather, short_own, long_rows):
    """Every line within reach of the first or the last sample of the grid, so that nearly every
    window is clipped by the reference (`minj = 0`, `maxj = dnwn`, _extcoeff.c:286-289).  The
    packed records of the staged gathers keep such windows unclipped (their consumers clamp to
    the tile; clipped, every group was a segment of its own and the edge tiles of a layer ran
    twice as long as the others) -- the sums and the zero pattern must not notice.
    short_own: the oversampled grid `own` stops 60 samples before `wn` does, so that
    ceil(dnwn / scale) < nwave and the upper clip is NOT redundant (it must then stay).
    long_rows: rows of several LDS chunks (the per-(group, chunk) record format)."""
    from pyratbay_amd import synth
    if long_rows:
        kw = dict(wnosamp=12, nlor=10, ndop=5, extent=4000.0, cutoff=80.0, niso=2)
        nwave, nlines, nl = 4097, 1500, 3
    else:
        kw = dict(wnosamp=24, nlor=18, ndop=9, extent=80.0, cutoff=3.0, niso=2)
        nwave, nlines, nl = 5001, 6000, 4
    case = synth.lbl_case(nwave, nl, nlines, seed=77, **kw)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    osamp = g['wnosamp']
    own = g['own'][:len(g['own']) - 60 * osamp] if short_own else g['own']
    # move the lines to within a few window half-widths of either end of `own` (isotope blocks
    # stay sorted by wavenumber)
    rng = np.random.default_rng(5)
    span = own[-1] - own[0]
    reach = min(0.02 * span, (vg['cutoff'] if vg['cutoff'] > 0 else 1.0) * 1.5)
    lwn = ln['lwn'].copy()
    lo = rng.random(len(lwn)) < 0.5
    lwn[lo] = own[0] + rng.random(lo.sum()) * reach
    lwn[~lo] = own[-1] - rng.random((~lo).sum()) * reach
    lid = ln['lid']
    order = np.lexsort((lwn, lid))
    lwn, lid = lwn[order], lid[order]
    elow, gf = ln['elow'][order], ln['gf'][order]
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], osamp)
    ll = eng.LineList(lwn, elow, gf, lid, len(iso['isomass']), own)
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=nl)
    lbl.set_gather_mode(gather)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    ext = host(lbl.extinction(t, d, z, add=True))
    profile = vt.flat()
    worst, touched = 0.0, 0
    for layer in range(nl):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], own, g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), iso['isoiext'],
                       lwn, elow, gf, lid, vg['cutoff'], 1e-30, atm['temp'][layer], 0, 1, 0)
        got = ext[layer]
        assert np.array_equal(got == 0, want == 0), (layer, 'zero pattern')
        nz = want != 0
        touched += int(nz.sum())
        worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL)
        assert want[0, 0] != 0 and (short_own or want[0, -1] != 0)     # the edges are reached
        if short_own:
            assert np.all(want[0, -59:] == 0)      # nothing beyond the oversampled grid
    # shards cut inside the edge tiles concatenate exactly
    parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
             for a, b in ((0, 37), (37, nwave - 45), (nwave - 45, nwave))]
    assert np.array_equal(np.concatenate(parts, axis=2), ext)
    print(f'edge windows {gather} short_own={short_own} long_rows={long_rows}: {touched} samples, '
          f'max rel err vs oracle = {worst:.2e}')


@pytest.mark.parametrize('gather', ['staged', 'global'])
def test_window_map_of_two_phase_shards(eng, monkeypatch, gather):
    """k_records of a two-phase shard call walks only the groups within reach of the shard (the
    window map: runs of in-window groups per (isotope, phase) resp. per isotope, thread -> group by
    a bisection over the run offsets).  With the map switched off (a thread per group of the whole
    list, out-of-window ones returning at once) the records, the local maxima and the sums must be
    the same bit for bit -- also when a plan is reused for another shard (the map is cached per
    window) and for a shard that no group reaches."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.setenv('PB_STAGE_SPLIT', '1')
    case = synth.lbl_case(9001, 4, 30000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=3, seed=9)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    # no lines in the last tenth of the grid: a shard there has an empty window
    keep = ln['lwn'] < g['wn'][0] + 0.85 * (g['wn'][-1] - g['wn'][0])
    lwn, elow, gf, lid = (ln[k][keep] for k in ('lwn', 'elow', 'gf', 'lid'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(lwn, elow, gf, lid, 3, g['own'])
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def shard_results(with_map):
        if with_map:
            monkeypatch.delenv('PB_NO_WINDOW_MAP', raising=False)
        else:
            monkeypatch.setenv('PB_NO_WINDOW_MAP', '1')
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], 1e-30, max_layers=4)
        lbl.set_gather_mode(gather)
        out = []
        for a, b in ((0, 2000), (2000, 5000), (0, 2000), (8700, 9001), (5000, 8700)):
            ext = lbl.extinction_begin(t, d, z, add=True, wbegin=a, wcount=b - a)
            kmax = lbl.kmax_tensor().clone()
            lbl.extinction_end()
            out.append((kmax, ext.clone()))
        return out

    on, off = shard_results(True), shard_results(False)
    for (k1, e1), (k0, e0) in zip(on, off):
        assert torch.equal(k1, k0)
        assert torch.equal(e1, e0)
    assert torch.count_nonzero(on[3][1]) == 0          # the empty shard
    assert torch.count_nonzero(on[1][1]) > 0


@pytest.mark.gpu_experiments
def test_rounds_split_keeps_the_window_map(eng, monkeypatch):
    """Round 2 defect (ADVICE): the partial-sum reallocation of the `rounds` gather freed the
    window map of two-phase shard calls while the plan's cache still described it; the next
    begin() handed the dangling pointer to k_records.  Two begin/end calls on ONE plan with the
    round gather split three ways (the second call re-uses the cached map after the partial sums
    were allocated) must equal the one-call form bit for bit."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.setenv('PB_STAGE_SPLIT', '3')
    case = synth.lbl_case(9001, 4, 30000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=3, seed=10)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 3, g['own'])

    def plan():
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], 1e-30, max_layers=4)
        lbl.set_gather_mode('rounds')
        return lbl
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    a, b = 2000, 6500
    one = plan()
    want = one.extinction(t, d, z, add=True, wbegin=a, wcount=b - a).clone()
    two = plan()
    for trip in range(3):
        got = two.extinction_begin(t, d, z, add=True, wbegin=a, wcount=b - a)
        # a single "rank": its own maxima are NOT the global ones, so take them from the
        # one-call plan (what the all-reduce would deliver)
        two.kmax_tensor().copy_(one.kmax_tensor())
        two.extinction_end()
        assert torch.equal(got, want), trip


@pytest.mark.parametrize('gather', ['staged', 'global'])
def test_window_map_too_large_for_lds(eng, monkeypatch, gather):
    """The run offsets of the phase-order window map are niso * osamp + 1 words: 16 isotopes at
    wnosamp 840 are 54 KB, more than k_records keeps in LDS (nothing capped the
    size; 20 isotopes at the reference's default wnosamp of 2160 exceeded the CU's 160 KB and the
    launch failed).  Such maps are bisected in global memory: same records, maxima and sums as
    with the map switched off, and as with the LDS form forced off on a small map."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.setenv('PB_STAGE_SPLIT', '1')
    sp = ('H2', 'He', 'H2O', 'CO', 'CO2', 'CH4')
    case = synth.lbl_case(1201, 3, 1500, wnosamp=840, nlor=6, ndop=4, extent=20.0, cutoff=1.0,
                          niso=4, seed=11, species=sp, vmr=(0.85, 0.149, 4e-4, 5e-4, 1e-7, 1e-4),
                          line_species=('H2O', 'CO', 'CO2', 'CH4'))
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    niso = len(iso['isomass'])
    assert niso * g['wnosamp'] * 4 > 48 * 1024
    iso['isoiext'] = np.zeros(niso, np.int32)
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], g['wnosamp'])
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def shard_results(env):
        for k in ('PB_NO_WINDOW_MAP', 'PB_WM_LDS_CAP'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], 1e-30, max_layers=3)
        lbl.set_gather_mode(gather)
        out = []
        for a, b in ((0, 400), (400, 1000), (1000, 1201)):
            ext = lbl.extinction_begin(t, d, z, add=True, wbegin=a, wcount=b - a)
            kmax = lbl.kmax_tensor().clone()
            lbl.extinction_end()
            out.append((kmax, ext.clone()))
        return out
    big, off, huge_cap = (shard_results({}), shard_results({'PB_NO_WINDOW_MAP': '1'}),
                          shard_results({'PB_WM_LDS_CAP': '65000'}))
    for (k1, e1), (k0, e0), (k2, e2) in zip(big, off, huge_cap):
        assert torch.equal(k1, k0) and torch.equal(e1, e0)
        assert torch.equal(k2, k0) and torch.equal(e2, e0)
    assert torch.count_nonzero(big[1][1]) > 0


def test_empty_shard_contributes_zero_maxima(eng):
    """An empty wavenumber shard (more ranks than samples, or a caller's uneven split) still
    takes part in the all-reduce(MAX) of the per-row maxima: begin() must leave zeros there, not
    the maxima of the plan's previous, hotter call."""
    import torch
    from pyratbay_amd import synth
    case = synth.lbl_case(3001, 3, 4000, wnosamp=24, nlor=12, ndop=6, extent=60.0, cutoff=3.0,
                          niso=2, seed=12)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=3)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    lbl.extinction_begin(t, d, z, add=True, wbegin=0, wcount=3001)
    assert torch.count_nonzero(lbl.kmax_tensor()) > 0
    lbl.extinction_end()
    lbl.extinction_begin(t, d, z, add=True, wbegin=3001, wcount=0,
                         out=torch.empty((3, 1, 0), dtype=torch.float64, device='cuda'))
    assert torch.count_nonzero(lbl.kmax_tensor()) == 0
    lbl.extinction_end()


@pytest.mark.parametrize('long_rows', [False, True])
@pytest.mark.parametrize('ethresh', [1e-30, 1e-3])
def test_chunked_line_list_equals_one_call(eng, orc, monkeypatch, long_rows, ethresh):
    """Out-of-core line lists (pb_lbl_set_record_budget): when the per-(layer, group) records
    exceed the budget, the phase-sorted group list is walked in chunks -- one pass for the
    per-row maxima of ALL lines, then per chunk its records and a gather that continues the
    running sums.  Same terms, same order: bit-identical to the one-call form with one
    workgroup per tile, for add = 1 and one row per species, with a threshold that drops most
    lines (the maxima must be the global ones), on a wavenumber shard, with rows of several LDS
    chunks; and equal to the oracle."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.setenv('PB_STAGE_SPLIT', '1')
    if long_rows:
        kw = dict(wnosamp=12, nlor=10, ndop=5, extent=4000.0, cutoff=80.0, niso=2)
        nwave, nlines, nl = 4097, 3000, 3
    else:
        kw = dict(wnosamp=24, nlor=18, ndop=9, extent=80.0, cutoff=3.0, niso=3)
        nwave, nlines, nl = 9001, 40000, 5
    case = synth.lbl_case(nwave, nl, nlines, seed=31, **kw)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    niso = len(iso['isomass'])
    iso['isoiext'] = np.array([0, 1, 0][:niso], np.int32)
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], g['wnosamp'])
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])

    def plan():
        lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                      iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                      vg['cutoff'], ethresh, max_layers=nl)
        lbl.set_gather_mode('staged')
        return lbl
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    whole, cut = plan(), plan()
    total = ll.ngroups * nl * 16
    for add in (True, False):
        want = whole.extinction(t, d, z, add=add).clone()
        assert whole.last_chunks == 0
        for parts in (2.5, 7.3):
            cut.set_record_budget(int(total / parts))
            got = cut.extinction(t, d, z, add=add)
            assert cut.last_chunks >= int(parts) + 1
            assert torch.equal(got, want), (add, parts)
        a, b = nwave // 3, nwave // 3 + 2500
        part = cut.extinction(t, d, z, add=add, wbegin=a, wcount=b - a)
        assert torch.equal(part, want[:, :, a:b])
    if ethresh > 1e-10:
        loose = plan()
        loose.set_ethresh(1e-30)
        assert not torch.equal(loose.extinction(t, d, z, add=True),
                               whole.extinction(t, d, z, add=True))
    # the chunked result against the oracle (one layer)
    cut.set_record_budget(int(total / 4.1))
    got = host(cut.extinction(t, d, z, add=True))[1]
    profile = vt.flat()
    want = np.zeros((1, g['nwave']))
    orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                   g['own'], g['divisors'], atm['dens'][1], atm['mol_radius'], atm['mol_mass'],
                   iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoz'][:, 1].copy(),
                   iso['isoiext'], ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'],
                   ethresh, atm['temp'][1], 0, 1, 0)
    assert np.array_equal(got == 0, want == 0)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    # a budget below one (isotope, phase) key's records cannot be met; two-phase shard calls and
    # the round gather are not chunked: loud errors, not silent over-allocation
    cut.set_record_budget(16 * nl)
    with pytest.raises(Exception, match='record budget'):
        cut.extinction(t, d, z, add=True)
    cut.set_record_budget(int(total / 3))
    with pytest.raises(Exception, match='record budget'):
        cut.extinction_begin(t, d, z, add=True, wbegin=0, wcount=2000)


@pytest.mark.parametrize('deep', ['0.5,3', '0.3,2', '1.0,8'])
def test_per_layer_phase_split(eng, orc, monkeypatch, deep):
    """The deepest layers of a staged launch can be cut into more pieces than the others (a unit
    table instead of one split for every layer; their partial sums are added in a fixed order by
    k_combine_layer_parts).  Same terms: equal to the unsplit launch to 1e-13 and to the oracle,
    identical zero pattern, two runs bitwise equal -- also with every layer deep and with the
    base launch already split."""
    import torch
    from pyratbay_amd import synth
    case = synth.lbl_case(9001, 11, 40000, wnosamp=24, nlor=18, ndop=9, extent=80.0,
                          cutoff=3.0, niso=3, seed=41)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 3, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=11)
    lbl.set_gather_mode('staged')
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    monkeypatch.setenv('PB_STAGE_DEEP', '0')
    plain = lbl.extinction(t, d, z, add=True).clone()
    monkeypatch.setenv('PB_STAGE_DEEP', deep)
    got = lbl.extinction(t, d, z, add=True).clone()
    assert torch.equal(lbl.extinction(t, d, z, add=True), got)
    assert torch.equal(got == 0, plain == 0)
    rel = ((got - plain).abs() / plain.abs().clamp_min(1e-300)).max().item()
    assert rel <= 1e-13, rel
    rows = lbl.extinction(t, d, z, add=False)                    # two output rows, same split
    monkeypatch.setenv('PB_STAGE_DEEP', '0')
    rows0 = lbl.extinction(t, d, z, add=False)
    assert ((rows - rows0).abs() / rows0.abs().clamp_min(1e-300)).max().item() <= 1e-13
    profile = vt.flat()
    for layer in (0, 5, 10):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                       ln['gf'], ln['lid'], vg['cutoff'], 1e-30, atm['temp'][layer], 0, 1, 0)
        np.testing.assert_allclose(host(got)[layer], want, rtol=RTOL)


def _resolution_case(seed=23, nlayers=8, nlines=6000):
    from pyratbay_amd import synth
    return synth.lbl_case(3001, nlayers, nlines, wnosamp=24, nlor=16, ndop=8, extent=60.0,
                          cutoff=3.0, niso=2, seed=seed, resolution=50000.0)


def _resolution_plan(eng, case, **kw):
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'], 2)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], len(iso['isomass']), g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'],
                  case['ethresh'], resolution=True, max_layers=atm['nlayers'], **kw)
    return vt, ll, lbl


def test_resolution_dynamic_grids(eng, monkeypatch):
    """`resolution` plans, gather mode 'dynamic' (one constant-step sub-plan per oversampling
    factor, pbhip.h: pb_lbl_set_gather_mode) against the direct gather of the same plan: every
    layer to 1e-12 with the same zero pattern, both add modes; the per-row maxima and factors
    reported by the plan are the direct call's; a wavenumber shard equals the slice of the whole
    call bit for bit; one side stream or four, the same bits; layers in an order that brings a
    factor back after another one (a temperature inversion) give the permuted result; the
    re-cut tables belong to the Voigt table and are shared by a second plan."""
    import torch
    case = _resolution_case()
    atm, iso = case['atm'], case['iso']
    nl = atm['nlayers']
    vt, ll, lbl = _resolution_plan(eng, case)
    bytes0 = vt.device_bytes
    temp, dens, isoz = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    for add in (True, False):
        lbl.set_gather_mode('auto')
        want = lbl.extinction(temp, dens, isoz, add=add)
        assert lbl.last_gather_kernel == 'k_ext_linterp'
        of0, kmax0 = lbl.last_state(nl, want.shape[1])
        lbl.set_gather_mode('dynamic')
        got = lbl.extinction(temp, dens, isoz, add=add)
        assert lbl.last_gather_kernel == 'dynamic grids'
        of1, kmax1 = lbl.last_state(nl, want.shape[1])
        assert np.array_equal(of0, of1) and np.array_equal(kmax0, kmax1)
        w, g_ = host(want), host(got)
        assert np.array_equal(w == 0, g_ == 0)
        np.testing.assert_allclose(g_, w, rtol=1e-12)
        assert np.count_nonzero(w) > 0.3 * w.size
    factors = len(set(of1.tolist()))
    assert factors >= 3, of1                               # several sub-plans were exercised
    assert vt.device_bytes > bytes0                        # the re-cut tables are accounted for
    whole = lbl.extinction(temp, dens, isoz, add=True)
    # a shard: the sub-plans compute the dynamic samples its outputs read, nothing else
    w0, wc = 700, 1111
    shard = lbl.extinction(temp, dens, isoz, add=True, wbegin=w0, wcount=wc)
    assert torch.equal(shard, whole[:, :, w0:w0 + wc])
    # accumulation: a second call into the same array doubles it
    twice = lbl.extinction(temp, dens, isoz, add=True, out=whole.clone())
    assert torch.equal(twice, whole + whole)
    monkeypatch.setenv('PB_RES_DYN_STREAMS', '1')
    assert torch.equal(lbl.extinction(temp, dens, isoz, add=True), whole)
    monkeypatch.delenv('PB_RES_DYN_STREAMS')
    # layers shuffled: equal factors are no longer neighbours
    perm = torch.tensor([0, 4, 1, 5, 2, 6, 3, 7][:nl], device='cuda')
    shuffled = lbl.extinction(temp[perm].contiguous(), dens[perm].contiguous(),
                              isoz[:, perm].contiguous(), add=True)
    assert torch.equal(shuffled, whole[perm])
    # species flags and threshold reach the sub-plans
    flags = iso['isoiext'].copy()
    flags[1] = -1
    for mode in ('auto', 'dynamic'):
        lbl.set_gather_mode(mode)
        lbl.set_isoiext(flags)
        lbl.set_ethresh(1e-3)
        out = host(lbl.extinction(temp, dens, isoz, add=True))
        if mode == 'auto':
            ref = out
    assert np.array_equal(ref == 0, out == 0)
    np.testing.assert_allclose(out, ref, rtol=1e-12)
    assert not np.array_equal(ref, host(whole))
    # a second plan on the same table re-uses the re-cut copies
    before = vt.device_bytes
    lbl2 = eng.LBL(vt, ll, case['grid']['wn'], case['grid']['divisors'], atm['mol_radius'],
                   atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                   iso['isoiext'], case['voigt']['cutoff'], case['ethresh'], resolution=True,
                   max_layers=nl)
    lbl2.set_gather_mode('dynamic')
    assert torch.equal(lbl2.extinction(temp, dens, isoz, add=True), whole)
    assert vt.device_bytes == before
    with pytest.raises(Exception, match='resolution'):
        build(eng, cases.extinction_inputs(), False, False, None)[2].set_gather_mode('dynamic')


@pytest.mark.parametrize('gather', ['auto', 'dynamic'])
def test_wavelength_step_grid(eng, orc, gather):
    """The reference's `wlstep` mode is the same code path as `resolution` (an output grid that
    is not a constant wavenumber step, _extcoeff.c:320-326): a constant-WAVELENGTH-step grid,
    direct gather and dynamic grids, every layer against the oracle."""
    from pyratbay_amd import synth
    case = synth.lbl_case(2501, 6, 5000, wnosamp=24, nlor=14, ndop=7, extent=60.0, cutoff=3.0,
                          niso=2, seed=31, resolution=40000.0)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    # same fine grid, outputs at a constant wavelength step between its ends (ascending wn)
    wl = np.linspace(1.0 / g['own'][-1], 1.0 / g['own'][0], 1800)
    wn = (1.0 / wl)[::-1].copy()
    wn[0] = g['own'][0]                                    # the package keeps wn[0] == own[0]
    wn = wn[wn <= g['own'][-1]]
    assert abs((1 / wn[3] - 1 / wn[4]) / (1 / wn[1000] - 1 / wn[1001]) - 1) < 1e-9
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'], 2)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    lbl = eng.LBL(vt, ll, wn, g['divisors'], atm['mol_radius'], atm['mol_mass'], iso['isoimol'],
                  iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'], 1e-30,
                  resolution=True, max_layers=6)
    lbl.set_gather_mode(gather)
    got = host(lbl.extinction(eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])))
    profile = vt.flat()
    for layer in range(atm['nlayers']):
        want = np.zeros((1, len(wn)))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], wn,
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                       ln['gf'], ln['lid'], vg['cutoff'], 1e-30, atm['temp'][layer], 0, 1, 1)
        assert np.array_equal(got[layer] == 0, want == 0)
        np.testing.assert_allclose(got[layer], want, rtol=RTOL)
    assert np.count_nonzero(got) > 0.3 * got.size


@pytest.mark.gpu_experiments
@pytest.mark.parametrize('wnosamp,cutoff,extent,niso,nwave,nlines', [
    (24, 9.0, 300.0, 2, 9001, 30000),       # cutoff-limited windows of ~360 samples: 7-chunk visits
    (12, 3.0, 80.0, 4, 6500, 20000),        # short rows (1-2 DMA pieces), four isotopes
    (36, 6.0, 300.0, 3, 4099, 6000),        # sparse: most phase rows have one record or none
    (300, 2.5, 120.0, 1, 5000, 40000),      # more phases than 4 x 64: several phase rounds per wavefront
])
def test_wave_kernel_vs_oracle(eng, orc, monkeypatch, wnosamp, cutoff, extent, niso, nwave, nlines):
    """pb_wave.hip: the layers of short phase rows (<= 384 samples) through the wave-autonomous
    kernel, the others through the staged one in the same call.  Against the oracle on every layer
    (identical zero pattern, 1e-10), against the staged kernel alone (1e-12: the same terms in
    another association), bitwise reproducible, shards cut inside tiles and at tile edges
    concatenate bit for bit, un-added rows (add=0), and a phase split (small launch) too."""
    from pyratbay_amd import synth
    case = synth.lbl_case(nwave, 8, nlines, wnosamp=wnosamp, nlor=20, ndop=10, extent=extent,
                          cutoff=cutoff, niso=niso, seed=nwave + wnosamp)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], wnosamp)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    isoiext = np.arange(niso, dtype=np.int32) % 2          # two rows when not added
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext,
                  vg['cutoff'], 1e-30, max_layers=8)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    lbl.set_gather_mode('wave')
    ext = host(lbl.extinction(t, d, z, add=True))
    assert lbl.last_gather_kernel == 'k_ext_wave+k_ext_staged'
    wave = lbl.last_wave_layers(8)
    assert wave.sum() >= 2, wave                           # (the upper layers at least)
    assert np.array_equal(host(lbl.extinction(t, d, z, add=True)), ext)
    profile = vt.flat()
    worst = 0.0
    for layer in range(8):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], g['own'], g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), isoiext,
                       ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], 1e-30,
                       atm['temp'][layer], 0, 1, 0)
        got = ext[layer]
        assert np.array_equal(got == 0, want == 0), (layer, int(wave[layer]))
        nz = want != 0
        if nz.any():
            worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=f'layer {layer} wave={wave[layer]}')
    print(f'wave kernel osamp={wnosamp}: layers {wave.tolist()}, max rel err vs oracle = {worst:.2e}')
    # shards: inside a tile, at a tile edge (2048), a one-sample shard
    bounds = [0, 377, 2048, 2049, min(4100, nwave - 1), nwave]
    parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
             for a, b in zip(bounds[:-1], bounds[1:])]
    assert np.array_equal(np.concatenate(parts, axis=2), ext)
    # un-added rows against the staged kernel alone; their density-weighted sum is the added form
    rows = host(lbl.extinction(t, d, z, add=False))
    lbl.set_gather_mode('staged')
    rows_s = host(lbl.extinction(t, d, z, add=False))
    ext_s = host(lbl.extinction(t, d, z, add=True))
    assert lbl.last_gather_kernel == 'k_ext_staged'
    assert np.array_equal(rows == 0, rows_s == 0) and np.array_equal(ext == 0, ext_s == 0)
    np.testing.assert_allclose(rows, rows_s, rtol=1e-12)
    np.testing.assert_allclose(ext, ext_s, rtol=1e-12)
    # a pinned phase split (what small launches and multi-GPU shards use): same terms again
    lbl.set_gather_mode('wave')
    monkeypatch.setenv('PB_STAGE_SPLIT', '3')
    split = host(lbl.extinction(t, d, z, add=True))
    np.testing.assert_allclose(split, ext, rtol=1e-12)
    assert np.array_equal(split == 0, ext == 0)
    parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
             for a, b in zip(bounds[:-1], bounds[1:])]
    assert np.array_equal(np.concatenate(parts, axis=2), split)


@pytest.mark.parametrize('gather', cases.gathers('auto', 'global', 'staged', 'rounds', 'wave'))
def test_band_structured_list_vs_oracle(eng, orc, gather):
    """A line list with band heads (synth.band_positions: peak line density 300 x the
    background's, 2 % of the band lines at exactly another line's wavenumber) instead of uniform
    positions: dense tiles next to nearly empty ones, long co-add groups, equal sort keys.  Every
    layer against the oracle; shards cut through a band head concatenate bit for bit."""
    from pyratbay_amd import synth
    case = synth.lbl_case(12001, 6, 60000, wnosamp=24, nlor=18, ndop=9, extent=80.0, cutoff=3.0,
                          niso=2, seed=91, bands=dict(nbands=3, contrast=300.0))
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    w = ln['lwn'][ln['lid'] == 0]
    hist, _ = np.histogram(w, bins=200)
    assert hist.max() > 50 * max(np.median(hist), 1) and np.sum(np.diff(w) == 0) > 100
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 2, g['own'])
    assert ll.nadd > 1000                                  # co-added lines at the heads
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=6)
    lbl.set_gather_mode(gather)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    ext = host(lbl.extinction(t, d, z, add=True))
    profile = vt.flat()
    worst = 0.0
    for layer in range(6):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'],
                       g['wn'], g['own'], g['divisors'], atm['dens'][layer],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, layer].copy(), iso['isoiext'],
                       ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], 1e-30,
                       atm['temp'][layer], 0, 1, 0)
        got = ext[layer]
        assert np.array_equal(got == 0, want == 0), layer
        nz = want != 0
        worst = max(worst, np.max(np.abs(got[nz] / want[nz] - 1)))
        np.testing.assert_allclose(got, want, rtol=RTOL)
    print(f'band heads {gather}: max rel err vs oracle = {worst:.2e}, co-added {ll.nadd}')
    if gather != 'auto':
        head = int(np.argmax(np.histogram(w, bins=g['nwave'], range=(g['wn'][0], g['wn'][-1]))[0]))
        bounds = sorted({0, max(1, head - 3), min(g['nwave'] - 1, head + 700), g['nwave']})
        parts = [host(lbl.extinction(t, d, z, wbegin=a, wcount=b - a))
                 for a, b in zip(bounds[:-1], bounds[1:])]
        assert np.array_equal(np.concatenate(parts, axis=2), ext)


def test_mode_switch_on_one_plan_with_resident_layers(eng, orc):
    """Found by tools/fuzz_r4.py (round 4; the bug was in the round-3 library too): a plan used with
    the staged kernel (phase split > 1: partial-sum planes written) and THEN in automatic mode,
    where the resident-profile kernel computes layers the staged kernel skips, had the combine
    pass add the earlier call's planes to the resident kernel's result -- up to 2 x the truth on
    a used plan, correct on a fresh one (fresh zero pages).  The combine passes now leave the
    layers of another kernel alone."""
    from pyratbay_amd import synth
    case = synth.lbl_case(40001, 12, 150000, wnstep=0.02, wnosamp=60, nlor=12, ndop=6,
                          extent=60.0, cutoff=0.768, niso=3, seed=41013,
                          bands=dict(nbands=5, contrast=450.0, in_bands=0.84))
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 60)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 3, g['own'])
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def plan():
        return eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                       iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                       vg['cutoff'], 1e-6, max_layers=12)
    fresh = plan()
    want = host(fresh.extinction(t, d, z))
    assert fresh.last_gather_kernel == 'k_ext_resident+k_ext_staged'
    assert fresh.last_layer_kinds(12)[0].sum() >= 1            # some layers resident
    used = plan()
    for mode in cases.live('staged', 'auto', 'wave', 'auto', 'global', 'auto'):
        used.set_gather_mode(mode)
        got = host(used.extinction(t, d, z))
        assert np.array_equal(got == 0, want == 0), mode
        np.testing.assert_allclose(got, want, rtol=1e-12, err_msg=mode)
        if mode == 'auto':
            assert np.array_equal(got, want), 'automatic mode on a used plan != a fresh plan'
    profile = vt.flat()
    for layer in (0, 11):
        ref = np.zeros((1, g['nwave']))
        orc.extinction(ref, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                       ln['gf'], ln['lid'], vg['cutoff'], 1e-6, atm['temp'][layer], 0, 1, 0)
        np.testing.assert_allclose(want[layer], ref, rtol=RTOL)
