"""Whole hot path on the device (extinction -> optical depth -> spectrum) against the
oracle driven through the reference's call sequence, transit and emission, plus the
wavenumber-sharded form and band integration.  Needs an MI355X.

Tolerance: rtol 1e-10 on the spectrum (north_star asks <= 1e-6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


@pytest.fixture(scope='module')
def case():
    from pyratbay_amd import synth
    return synth.lbl_case(3001, 14, 12000, wnosamp=24, nlor=20, ndop=10, extent=80.0,
                          cutoff=4.0, niso=2, seed=11)


@pytest.fixture(scope='module')
def oracle_ec(case, orc):
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    size = vg['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, vg['lorentz'], vg['doppler'], g['ownstep'])
    ec = np.zeros((atm['nlayers'], g['nwave']))
    for layer in range(atm['nlayers']):
        row = np.zeros((1, g['nwave']))
        orc.extinction(row, profile, size, index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                       ln['gf'], ln['lid'], vg['cutoff'], case['ethresh'],
                       atm['temp'][layer], 0, 1, 0)
        ec[layer] = row[0]
    return ec


def test_transit_pipeline(eng, case, orc, oracle_ec):
    atm = case['atm']
    model = eng.LBLSpectrum(case, rt_path='transit')
    spectrum = model.run().cpu().numpy()
    np.testing.assert_allclose(model.ec.cpu().numpy()[:, 0], oracle_ec, rtol=RTOL)
    depth, ideep = orc.optical_depth_transit(oracle_ec, atm['radius'], 0, atm['nlayers'],
                                             case['maxdepth'])
    assert np.array_equal(model.ideep.cpu().numpy(), ideep)
    want = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    np.testing.assert_allclose(spectrum, want, rtol=RTOL)
    # the modulation is there (not a flat, clear-atmosphere spectrum)
    assert want.max() / want.min() > 1.0005


def test_emission_pipeline(eng, case, orc, oracle_ec):
    g, atm = case['grid'], case['atm']
    model = eng.LBLSpectrum(case, rt_path='emission')
    flux = model.run().cpu().numpy()
    nl, nw = atm['nlayers'], g['nwave']
    depth = np.zeros((nl, nw))
    ideep = np.full(nw, nl - 1, np.int32)
    orc.plane_parallel_optical_depth(depth, ideep, oracle_ec, -orc.ediff(atm['radius']),
                                     case['maxdepth'], 0, nl)
    B = orc.blackbody_wn_2D(g['wn'], atm['temp'])
    mu = model.mu.cpu().numpy()
    w = model.weights.cpu().numpy()
    inten = orc.intensity(depth, ideep, B, mu, 0)
    want = np.sum(inten * w[:, None], axis=0)
    assert np.array_equal(model.ideep.cpu().numpy(), ideep)
    np.testing.assert_allclose(flux, want, rtol=RTOL)


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_pipeline_equals_single(eng, case, world):
    """Each 'rank' computes its shard of the global grid; concatenation is bit-identical
    to the one-GPU spectrum (no halo exchange needed: SURVEY.md section 8e)."""
    from pyratbay_amd.dist import shard_bounds
    single = eng.LBLSpectrum(case, rt_path='transit')
    want = single.run().cpu().numpy()
    b = shard_bounds(case['grid']['nwave'], world)
    parts = []
    for r in range(world):
        m = eng.LBLSpectrum(case, rt_path='transit', wbegin=int(b[r]),
                            wcount=int(b[r + 1] - b[r]), voigt=single.voigt,
                            lines=single.lines)
        parts.append(m.run().cpu().numpy())
    assert np.array_equal(np.concatenate(parts), want)


def test_band_integration(eng, case):
    """PassBand.integrate (spec_tools.py:193-233) on the device, whole and in shards."""
    import torch
    from pyratbay_amd.dist import shard_bounds
    wn = case['grid']['wn']
    rng = np.random.default_rng(3)
    spectrum = 0.01 + 1e-4 * rng.uniform(size=len(wn))
    bands = []
    for lo, hi in ((10, 700), (650, 2200), (0, len(wn)), (2999, 3001)):
        x = np.linspace(-1, 1, hi - lo)
        resp = np.exp(-3 * x**2) * (1e4 / wn[lo:hi])       # photon counting: wl folded in
        bands.append((lo, resp, 1.0 / np.trapezoid(resp, wn[lo:hi])))
    want = np.array([np.trapezoid(spectrum[s:s + len(r)] * r, wn[s:s + len(r)]) * h
                     for s, r, h in bands])
    pb = eng.PassBands(wn, bands)
    spec_d = eng.dev(spectrum)
    got = (pb.partial_integrate(spec_d) * pb.heights).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-13)
    b = shard_bounds(len(wn), 4)
    total = torch.zeros(len(bands), dtype=torch.float64, device='cuda')
    for r in range(4):
        total += pb.partial_integrate(spec_d, int(b[r]), int(b[r + 1] - b[r]))
    np.testing.assert_allclose((total * pb.heights).cpu().numpy(), want, rtol=1e-13)


def test_set_atmosphere_changes_result(eng, case):
    """eval()-style update of T / densities / partition functions without re-uploading
    lines or the Voigt table (pyrat_obj.py:258-300)."""
    from pyratbay_amd import synth
    model = eng.LBLSpectrum(case, rt_path='transit')
    base = model.run().cpu().numpy().copy()
    atm, iso = case['atm'], case['iso']
    temp = atm['temp'] * 1.1
    dens = atm['dens'] / 1.1
    isoz = synth.partition_function(temp)[None, :].repeat(len(iso['isomass']), 0)
    model.set_atmosphere(temp, dens, isoz)
    hot = model.run().cpu().numpy()
    assert np.max(np.abs(hot / base - 1)) > 1e-6
    model.set_atmosphere(atm['temp'], atm['dens'], iso['isoz'])
    assert np.array_equal(model.run().cpu().numpy(), base)


@pytest.mark.parametrize('world', [2, 8])
def test_layer_sharded_extinction_equals_single(eng, case, world):
    """Layer-sharded mode (dist.LayerShardedTransit): rank r computes the layers
    r, r+N, ...; interleaving the rows reproduces the single-GPU ec bit for bit, and the
    world=1 pipeline reproduces the single-GPU spectrum."""
    import torch
    from pyratbay_amd.dist import LayerShardedTransit
    single = eng.LBLSpectrum(case, rt_path='transit')
    want = single.run().cpu().numpy()
    ec_full = single.ec.cpu().numpy()[:, 0]
    nl = single.nlayers
    got = np.zeros_like(ec_full)
    for r in range(world):
        idx = torch.arange(r, nl, world, device='cuda')
        if len(idx) == 0:
            continue
        ec = single.lbl.extinction(single.temp[idx].contiguous(), single.dens[idx].contiguous(),
                                   single.isoz[:, idx].contiguous(), add=True)
        got[r::world] = ec.cpu().numpy()[:, 0]
    assert np.array_equal(got, ec_full)
    one = LayerShardedTransit(case, 1, 0)
    assert np.array_equal(one.step().cpu().numpy(), want)


def test_layer_sharded_step_in_resolution_mode(eng):
    """The `resolution` mode accumulates into the extinction array (like the reference), so the
    layer-sharded step zeroes its slice per spectrum: consecutive steps and submit()/flush()
    return the model's own spectrum again and again (they used to add up: found by a two-rank
    rehearsal of `bench.py --workload c2-res`); strided layer subsets through the dynamic grids
    interleave to the whole call's rows bit for bit."""
    import torch
    from pyratbay_amd import synth
    from pyratbay_amd.dist import LayerShardedTransit
    case = synth.lbl_case(2001, 7, 4000, wnosamp=24, nlor=14, ndop=7, extent=60.0, cutoff=3.0,
                          niso=2, seed=29, resolution=60000.0)
    single = eng.LBLSpectrum(case, rt_path='transit')
    want = single.run().clone()
    sh = LayerShardedTransit(case, 1, 0, voigt=single.voigt, lines=single.lines)
    for _ in range(3):
        assert torch.equal(sh.step(), want)
    got = [sh.submit() for _ in range(4)]
    got = [g for g in got if g is not None] + list(sh.flush())
    assert len(got) == 4 and all(torch.equal(g, want) for g in got)
    nl, ec_full = single.nlayers, single.ec.clone()
    for world in (2, 3):
        for r in range(world):
            idx = torch.arange(r, nl, world, device='cuda')
            ec = single.lbl.extinction(single.temp[idx].contiguous(),
                                       single.dens[idx].contiguous(),
                                       single.isoz[:, idx].contiguous(), add=True)
            assert single.lbl.last_gather_kernel == 'dynamic grids'
            assert torch.equal(ec, ec_full[idx])


def test_pipelined_submit_flush_single_process(eng, case):
    """The software-pipelined form of the layer-sharded step (submit/flush, double buffers,
    stages A/B/C of consecutive spectra interleaved) returns the same spectra as step(),
    in order, also when the atmosphere changes between submissions."""
    from pyratbay_amd import synth
    from pyratbay_amd.dist import LayerShardedTransit
    sh = LayerShardedTransit(case, 1, 0)
    want0 = sh.step().cpu().numpy().copy()
    atm, iso = case['atm'], case['iso']
    hot_t = atm['temp'] * 1.04
    hot_z = synth.partition_function(hot_t)[None, :].repeat(len(iso['isomass']), 0)

    def set_temp(temp, isoz):
        sh.temp.copy_(eng.dev(temp)[eng.dev(sh.layers).long()])
        sh.isoz.copy_(eng.dev(isoz)[:, eng.dev(sh.layers).long()])

    set_temp(hot_t, hot_z)
    want1 = sh.step().cpu().numpy().copy()
    assert not np.array_equal(want0, want1)
    set_temp(atm['temp'], iso['isoz'])
    got = []
    for k in range(5):
        set_temp(*((hot_t, hot_z) if k % 2 else (atm['temp'], iso['isoz'])))
        out = sh.submit()
        if out is not None:
            got.append(out.cpu().numpy().copy())
    got += [o.cpu().numpy().copy() for o in sh.flush()]
    assert len(got) == 5
    for k, g_ in enumerate(got):
        assert np.array_equal(g_, want1 if k % 2 else want0), k
    assert sh.flush() == []


def test_hip_graph_replay(eng, case):
    """The step captured as one HIP graph reproduces the eager spectrum, also after the
    atmosphere buffers were updated in place."""
    from pyratbay_amd import synth
    model = eng.LBLSpectrum(case, rt_path='transit')
    eager = model.run().cpu().numpy().copy()
    replay = model.capture()
    assert np.array_equal(replay().cpu().numpy(), eager)
    atm, iso = case['atm'], case['iso']
    temp = atm['temp'] * 1.05
    isoz = synth.partition_function(temp)[None, :].repeat(len(iso['isomass']), 0)
    model.set_atmosphere(temp, atm['dens'], isoz)
    hot = replay().cpu().numpy().copy()
    fresh = eng.LBLSpectrum(case, rt_path='transit', voigt=model.voigt, lines=model.lines)
    fresh.set_atmosphere(temp, atm['dens'], isoz)
    assert np.array_equal(hot, fresh.run().cpu().numpy())


def test_lbl_with_continuum(eng, case, orc):
    """Line-by-line extinction + continuum terms in the same ec, then the transit stages."""
    from pyratbay_amd import continuum as ct
    from oracle import continuum as oc
    g, atm = case['grid'], case['atm']
    wn, temp = g['wn'], atm['temp']
    pressure = np.logspace(-6, 2, atm['nlayers'])
    n_h2 = pressure * ct.BAR / (ct.K * temp) * 0.85
    lec = ct.Lecavelier(pressure, wn=wn)
    lec.calc_cross_section([1.0, -3.0])
    cont = ct.Continuum(wn, pressure, [ct.Kurucz(wn, 'H2'), lec])
    plain = eng.LBLSpectrum(case, rt_path='transit')
    plain.run()
    model = eng.LBLSpectrum(case, rt_path='transit', voigt=plain.voigt, lines=plain.lines,
                            continuum=cont, continuum_density={'H2': n_h2})
    spectrum = model.run().cpu().numpy()
    want_ec = (plain.ec.cpu().numpy()[:, 0]
               + oc.rayleigh_cross_section(wn, 'H2') * n_h2[:, None]
               + oc.lecavelier_cross_section(wn, [1.0, -3.0])
               * oc.nominal_density(pressure, temp)[:, None])
    np.testing.assert_allclose(model.ec.cpu().numpy()[:, 0], want_ec, rtol=1e-12)
    depth, ideep = orc.optical_depth_transit(want_ec, atm['radius'], 0, atm['nlayers'],
                                             case['maxdepth'])
    want = orc.transmission(depth, atm['radius'], float(atm['rstar']), ideep, 0)
    np.testing.assert_allclose(spectrum, want, rtol=1e-11)
    assert np.all(spectrum >= plain.spectrum.cpu().numpy())


def test_set_atmosphere_with_continuum(eng, case, orc):
    """A new atmosphere reaches EVERY opacity term: after set_atmosphere() the continuum is
    evaluated at the new temperatures and densities too (oracle: Rayleigh x n_H2 +
    Lecavelier x nominal density(p, T_new) on top of the LBL rows of the new atmosphere)."""
    from pyratbay_amd import continuum as ct, synth, _capi
    from oracle import continuum as oc
    g, atm, iso = case['grid'], case['atm'], case['iso']
    wn = g['wn']
    pressure = np.logspace(-6, 2, atm['nlayers'])
    lec = ct.Lecavelier(pressure, wn=wn)
    lec.calc_cross_section([1.0, -3.0])
    cont = ct.Continuum(wn, pressure, [ct.Kurucz(wn, 'H2'), lec])
    n_h2 = pressure * ct.BAR / (ct.K * atm['temp']) * 0.85
    model = eng.LBLSpectrum(case, rt_path='transit', continuum=cont,
                            continuum_density={'H2': n_h2})
    model.run()
    temp = atm['temp'] * 1.07
    dens = atm['dens'] / 1.07
    isoz = synth.partition_function(temp)[None, :].repeat(len(iso['isomass']), 0)
    n_h2_new = pressure * ct.BAR / (ct.K * temp) * 0.85
    with pytest.raises(_capi.PbError):
        model.set_atmosphere(temp, dens, isoz)          # continuum densities are required
    model.set_atmosphere(temp, dens, isoz, continuum_density={'H2': n_h2_new})
    spectrum = model.run().cpu().numpy()
    plain = eng.LBLSpectrum(case, rt_path='transit', voigt=model.voigt, lines=model.lines)
    plain.set_atmosphere(temp, dens, isoz)
    plain.run()
    want_ec = (plain.ec.cpu().numpy()[:, 0]
               + oc.rayleigh_cross_section(wn, 'H2') * n_h2_new[:, None]
               + oc.lecavelier_cross_section(wn, [1.0, -3.0])
               * oc.nominal_density(pressure, temp)[:, None])
    np.testing.assert_allclose(model.ec.cpu().numpy()[:, 0], want_ec, rtol=1e-12)
    depth, ideep = orc.optical_depth_transit(want_ec, atm['radius'], 0, atm['nlayers'],
                                             case['maxdepth'])
    want = orc.transmission(depth, atm['radius'], float(atm['rstar']), ideep, 0)
    np.testing.assert_allclose(spectrum, want, rtol=1e-11)


def test_layer_sharded_set_atmosphere(eng, case):
    """dist.LayerShardedTransit.set_atmosphere refreshes the rank's layer slices: two
    alternating atmospheres give what a fresh single-GPU model gives for each."""
    from pyratbay_amd import synth
    from pyratbay_amd.dist import LayerShardedTransit
    atm, iso = case['atm'], case['iso']
    sh = LayerShardedTransit(case, 1, 0)
    base = sh.step().cpu().numpy().copy()
    temp = atm['temp'] * 0.93
    dens = atm['dens'] / 0.93
    isoz = synth.partition_function(temp)[None, :].repeat(len(iso['isomass']), 0)
    ref = eng.LBLSpectrum(case, rt_path='transit', voigt=sh.model.voigt, lines=sh.model.lines)
    ref.set_atmosphere(temp, dens, isoz)
    want = ref.run().cpu().numpy()
    for _ in range(2):
        sh.set_atmosphere(temp, dens, isoz)
        assert np.array_equal(sh.step().cpu().numpy(), want)
        sh.set_atmosphere(atm['temp'], atm['dens'], iso['isoz'])
        assert np.array_equal(sh.step().cpu().numpy(), base)


@pytest.mark.parametrize('rt_path', ['transit', 'emission'])
def test_spectrum_pipeline_equals_serial_runs(eng, case, rt_path):
    """engine.SpectrumPipeline: consecutive spectra of alternating atmospheres in flight on two
    and three HIP streams -- every spectrum bit-identical to LBLSpectrum.run() of the same
    atmosphere, in submission order, with the shared Voigt table and line list untouched."""
    import torch
    from pyratbay_amd import synth
    atm, iso = case['atm'], case['iso']
    atmospheres = []
    for f in (1.0, 1.1, 0.93, 1.04, 0.97):
        temp = atm['temp'] * f
        isoz = synth.partition_function(temp)[None, :].repeat(len(iso['isomass']), 0)
        atmospheres.append((temp, atm['dens'] / f, isoz, atm['radius'] * (1.0 + 0.01 * (f - 1))))
    serial = eng.LBLSpectrum(case, rt_path=rt_path)
    want = []
    for a in atmospheres:
        serial.set_atmosphere(*a)
        want.append(serial.run().cpu().numpy().copy())
    assert np.max(np.abs(want[1] / want[0] - 1)) > 1e-6
    for depth in (2, 3):
        pipe = eng.SpectrumPipeline(case, depth=depth, rt_path=rt_path, voigt=serial.voigt,
                                    lines=serial.lines)
        assert pipe.models[1].voigt is serial.voigt and pipe.models[1].lines is serial.lines
        got = []
        for rep in range(2):                       # the contexts are reused
            pending = []
            for i, a in enumerate(atmospheres):
                out, event = pipe.submit(a)
                pending.append((out, event))
                if len(pending) == depth:          # consume before the context is reused
                    o, e = pending.pop(0)
                    e.synchronize()
                    got.append(o.cpu().numpy().copy())
            pipe.flush()
            torch.cuda.current_stream().synchronize()
            got.extend(o.cpu().numpy().copy() for o, _ in pending)
        assert len(got) == 2 * len(atmospheres)
        for i, g_ in enumerate(got):
            assert np.array_equal(g_, want[i % len(atmospheres)]), (depth, i)


@pytest.mark.parametrize('rt_path', ['transit', 'emission'])
def test_stage_timestamps_like_the_reference(eng, case, rt_path):
    """run() fills `timestamps` with the reference's keys -- 'extinction', 'odepth', 'spectrum'
    (pyrat_obj.py:203-214), seconds per stage -- from HIP events resolved when read.  Their sum
    is bracketed by the wall time of a synchronised run; switching the timers off changes no
    result bit; the transit geometry (ONE library call for optical depth + transmission) still
    reports the two stages separately."""
    import time
    import torch
    model = eng.LBLSpectrum(case, rt_path=rt_path)
    model.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    want = model.run().clone()
    ts = model.timestamps                      # waits for the run
    wall = time.perf_counter() - t0
    assert list(ts) == ['extinction', 'odepth', 'spectrum']
    assert all(v > 0.0 for v in ts.values())
    assert sum(ts.values()) <= wall * 1.05 + 1e-4
    assert ts['extinction'] > ts['spectrum'] * 0.2      # the line-by-line stage is not a no-op
    plain = eng.LBLSpectrum(case, rt_path=rt_path, voigt=model.voigt, lines=model.lines,
                            timestamps=False)
    assert torch.equal(plain.run(), want)
    with pytest.raises(Exception):
        plain.timestamps
    # a second run replaces the first one's stamps; a range can be opened around anything
    with eng.profiler_range('second run'):
        model.run()
    assert list(model.timestamps) == ['extinction', 'odepth', 'spectrum']


def test_finished_timer_ignores_later_calls_and_captures(eng, case):
    """a fused transit call marks its internal 'odepth' boundary on the timer
    the calling thread started last.  A run that has FINISHED must not collect the boundaries of
    later calls (a timestamps=False model, a direct engine.transit_spectrum): its values stay
    what they were.  And a run captured into a HIP graph records no timer events at all -- the
    timestamps of the last eager run remain readable after capture() and replay()."""
    import torch
    model = eng.LBLSpectrum(case, rt_path='transit')
    model.run()
    model.run()
    before = dict(model.timestamps)
    assert list(before) == ['extinction', 'odepth', 'spectrum']
    other = eng.LBLSpectrum(case, rt_path='transit', voigt=model.voigt, lines=model.lines,
                            timestamps=False)
    for _ in range(3):
        other.run()
    eng.transit_spectrum(model.ec.view(model.nlayers, model.wcount), model.raypath, model.radius,
                         model.rstar, model.itop, model.nlayers, model.maxdepth)
    torch.cuda.synchronize()
    assert dict(model.timestamps) == before
    want = model.run().clone()
    replay = model.capture()                    # one eager run, one warm-up, one captured run
    eager = dict(model.timestamps)
    assert list(eager) == ['extinction', 'odepth', 'spectrum'] and all(v > 0 for v in eager.values())
    out = replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert dict(model.timestamps) == eager


def test_stage_timer_api(eng):
    """The C-ABI timer on its own: stages in order, duplicates summed, errors for a timer that
    was not started or is full."""
    import torch
    t = eng.StageTimer(max_stages=3)
    with pytest.raises(Exception):
        t.mark('early')
    t.start('a')
    x = torch.ones(1 << 20, device='cuda')
    t.mark('a', 'b')
    x = (x * 2).sum()
    t.mark('b', 'a')
    t.mark('a')
    with pytest.raises(Exception):
        t.mark('overflow')
    got = t.read()
    assert list(got) == ['a', 'b'] and all(v >= 0.0 for v in got.values())
    t.start()
    assert t.read() == {}
    t.close()


def test_shard_pipeline_single_process(eng, case):
    """dist.ShardPipeline (the multi-GPU wavenumber form with several spectra in flight per
    rank) in its one-rank form: every submitted spectrum equals LBLSpectrum.run() of the same
    atmosphere; the concurrency hint changes the tiling of a launch, never a term (1e-12)."""
    import torch
    from pyratbay_amd.dist import ShardPipeline
    serial = eng.LBLSpectrum(case, rt_path='transit')
    want = serial.run().clone()
    pipe = ShardPipeline(case, 1, 0, depth=3, voigt=serial.voigt, lines=serial.lines)
    outs = []
    for i in range(5):
        r = pipe.submit()                  # (the spectrum submitted before this one, see submit())
        assert (r is None) == (i == 0)
        if r is not None:
            r[1].synchronize()
            outs.append(r[0].clone())
    last = pipe.flush()
    torch.cuda.synchronize()
    outs.append(last[0].clone())
    assert len(outs) == 5 and pipe.flush() is None
    for o in outs:
        np.testing.assert_allclose(o.cpu().numpy(), want.cpu().numpy(), rtol=1e-12)
    # a plan told about the other spectra in flight gives the same extinction
    hinted = eng.LBLSpectrum(case, rt_path='transit', voigt=serial.voigt, lines=serial.lines)
    hinted.lbl.set_concurrency(4)
    hinted.run()
    np.testing.assert_allclose(hinted.ec.cpu().numpy(), serial.ec.cpu().numpy(), rtol=1e-12)
    with pytest.raises(Exception):
        hinted.lbl.set_concurrency(0)


@pytest.mark.parametrize('stack', [2, 3])
def test_shard_pipeline_stacked_atmospheres(eng, case, monkeypatch, stack):
    """dist.ShardPipeline(stack=K): K atmospheres per extinction call (dist.StackedShard).  K
    DIFFERENT atmospheres in one submission each equal LBLSpectrum.run() of that atmosphere on
    the same shard (1e-12; bit for bit with the phase split pinned), over several submissions and
    through flush(); the two-phase form (kmax exchange hook) included."""
    import torch
    from pyratbay_amd.dist import ShardPipeline
    atm, iso = case['atm'], case['iso']
    nwave = case['grid']['nwave']
    a, b = nwave // 5, nwave // 5 + nwave // 3                # a shard in the middle
    serial = eng.LBLSpectrum(case, rt_path='transit', wbegin=a, wcount=b - a)
    atms = []
    for k in range(stack):
        temp = atm['temp'] * (1.0 + 0.04 * k) + 3.0 * k
        dens = atm['dens'] * (atm['temp'] / temp)[:, None]
        isoz = iso['isoz'] * (1.0 + 0.01 * k)
        radius = atm['radius'] * (1.0 + 0.002 * k)
        atms.append((temp, dens, isoz, radius))
    want = []
    for t in atms:
        serial.set_atmosphere(*t)
        want.append(serial.run().clone())
    assert not torch.equal(want[0], want[1])
    for pinned in (False, True):
        if pinned:
            monkeypatch.setenv('PB_STAGE_SPLIT', '2')
            want = []
            for t in atms:
                serial.set_atmosphere(*t)
                want.append(serial.run().clone())
        pipe = ShardPipeline(case, 1, 0, depth=2, voigt=serial.voigt, lines=serial.lines,
                             stack=stack)
        # (world 1: the pipeline's shard is the whole grid; compare on the serial model's shard)
        for m in pipe.models:
            m.kmax_exchange = lambda t: None                  # the two-phase form, one rank
            for k, t in enumerate(atms):
                m.set_atmosphere(k, *t)
        outs = []
        for i in range(3):
            r = pipe.submit()
            assert (r is None) == (i == 0)
            if r is not None:
                r[1].synchronize()
                outs.append([x.clone() for x in r[0]])
        last = pipe.flush()
        torch.cuda.synchronize()
        outs.append([x.clone() for x in last[0]])
        assert len(outs) == 3 and all(len(o) == stack for o in outs)
        for o in outs:
            for k in range(stack):
                got = o[k][a:b]
                if pinned:
                    assert torch.equal(got, want[k]), k
                np.testing.assert_allclose(got.cpu().numpy(), want[k].cpu().numpy(), rtol=1e-12)
        del pipe


@pytest.mark.parametrize('nlayers,stack', [(120, 4), (200, 3), (80, 8)])
def test_stacked_shard_many_layers(eng, nlayers, stack):
    """dist.StackedShard beyond the bench's shape: C4's 120 layers x 4 atmospheres, 600 and 640
    stacked layers in one extinction call (layer indices above 255 in the unit table, the per-row
    maxima of every stacked layer) -- each atmosphere's spectrum equals the serial run's."""
    import torch
    from pyratbay_amd import synth
    from pyratbay_amd.dist import StackedShard
    case = synth.lbl_case(2001, nlayers, 8000, wnosamp=24, nlor=20, ndop=10, extent=80.0,
                          cutoff=4.0, niso=2, seed=3)
    serial = eng.LBLSpectrum(case, rt_path='transit')
    atm, iso = case['atm'], case['iso']
    shard = StackedShard(case, stack, voigt=serial.voigt, lines=serial.lines)
    want = []
    for k in range(stack):
        temp = atm['temp'] * (1 + 0.03 * k)
        dens = atm['dens'] * (atm['temp'] / temp)[:, None]
        serial.set_atmosphere(temp, dens, iso['isoz'])
        want.append(serial.run().clone())
        shard.set_atmosphere(k, temp, dens, iso['isoz'])
    got = shard.run()
    torch.cuda.synchronize()
    assert not torch.equal(want[0], want[-1])
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.cpu().numpy(), w.cpu().numpy(), rtol=1e-12)


@pytest.mark.gpu_experiments
def test_resolution_mode_predicted_runs(eng, monkeypatch):
    """`resolution` mode with the run plan taken from the last read-back of the layers' factors
    (LBLSpectrum(predict_runs=True), pb_lbl_set_dyn_predict) instead of a stream synchronisation
    in every call.  A steady atmosphere: every spectrum bit for bit the default form's, the calls
    after the first planned from the prediction, none contradicted.  An atmosphere whose factors
    differ: the layers the plan does not fit go through the direct gather in the same call --
    right at once (1e-12 of a fresh default model, same zero pattern), the contradiction is
    noticed and the next calls synchronise, after which the prediction is used again.  Such a
    model can be captured into a graph, and the graph stays right when the atmosphere changes."""
    import torch
    from pyratbay_amd import synth
    monkeypatch.delenv('PB_RES_DYN_PREDICT', raising=False)
    case = synth.lbl_case(2001, 8, 5000, wnosamp=24, nlor=14, ndop=7, extent=60.0, cutoff=3.0,
                          niso=2, seed=19, resolution=60000.0)
    atm, iso = case['atm'], case['iso']
    ref = eng.LBLSpectrum(case, rt_path='transit')
    want = ref.run().clone()
    want_ec = ref.ec.clone()
    model = eng.LBLSpectrum(case, rt_path='transit', voigt=ref.voigt, lines=ref.lines,
                            predict_runs=True)
    for _ in range(6):
        assert torch.equal(model.run(), want)
        assert torch.equal(model.ec, want_ec)
    torch.cuda.synchronize()
    spec, sync, missed = model.lbl.dyn_stats()
    assert sync == 1 and spec == 5 and missed == 0, (spec, sync, missed)
    assert ref.lbl.dyn_stats()[0] == 0                     # the default form never predicts
    # another atmosphere: hotter and reversed in pressure order -> other factors per layer
    temp2 = atm['temp'][::-1].copy() * 1.3
    dens2 = atm['dens'][::-1].copy() * 0.05
    isoz2 = iso['isoz'][:, ::-1].copy()
    ref.set_atmosphere(temp2, dens2, isoz2)
    want2 = ref.run().clone()
    want2_ec = ref.ec.clone()
    of_a, _ = ref.lbl.last_state(atm['nlayers'], 1)
    model.set_atmosphere(temp2, dens2, isoz2)
    got2 = model.run().clone()                            # planned from the OLD atmosphere
    assert model.lbl.dyn_stats()[0] == 6
    e, w = model.ec.cpu().numpy(), want2_ec.cpu().numpy()
    assert np.array_equal(e == 0, w == 0)
    np.testing.assert_allclose(e, w, rtol=1e-12)
    np.testing.assert_allclose(got2.cpu().numpy(), want2.cpu().numpy(), rtol=1e-12)
    torch.cuda.synchronize()
    for _ in range(12):                                    # contradiction seen: synchronous calls,
        assert torch.equal(model.run(), want2)             # (exact again), then predictions again
        torch.cuda.synchronize()
    spec, sync, missed = model.lbl.dyn_stats()
    assert missed >= 1 and sync >= 1 + 8 and spec >= 6 + 2, (spec, sync, missed)
    # capture: the graph holds the plan of this atmosphere ...
    replay = model.capture()
    assert torch.equal(replay(), want2)
    # ... and stays right for the first one (layers it does not fit: direct gather in the graph)
    model.set_atmosphere(atm['temp'], atm['dens'], iso['isoz'])
    out = replay().clone()
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=1e-12)
    e, w = model.ec.cpu().numpy(), want_ec.cpu().numpy()
    assert np.array_equal(e == 0, w == 0)
    np.testing.assert_allclose(e, w, rtol=1e-12)
    # the default form still refuses to be captured
    with pytest.raises(RuntimeError, match='cannot be captured'):
        ref.capture()


def test_resolution_mode_spectrum(eng, orc):
    """A constant-resolving-power output grid through LBLSpectrum (the reference's `resolution`
    mode: the kept samples are interpolated from the dynamic grid, _extcoeff.c:320-326 /
    utils.h:139-163, and ACCUMULATED): every layer of ec against the oracle, two runs equal (the
    model zeroes ec itself), the table kept in the reference layout only (keep_flat = 2), and a
    constant-step plan on such a table refused."""
    import torch
    from pyratbay_amd import synth
    case = synth.lbl_case(2001, 6, 5000, wnosamp=24, nlor=14, ndop=7, extent=60.0, cutoff=3.0,
                          niso=2, seed=17, resolution=60000.0)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    assert g['resolution'] == 60000.0 and abs(g['wn'][1] / g['wn'][0] - 1 - 1 / 60000.0) < 1e-9
    model = eng.LBLSpectrum(case, rt_path='transit')
    vt = model.voigt
    assert vt.device_bytes < 1.2 * 8 * vt.nprofile         # one layout on the device, not two
    first = model.run().clone()
    # (the model asks for the per-layer dynamic grids; a bare plan keeps the direct gather)
    assert model.lbl.last_gather_kernel == 'dynamic grids'
    assert torch.equal(model.run(), first)                 # ec is zeroed per run, not summed up
    model.lbl.set_gather_mode('auto')
    direct = model.run().clone()
    assert model.lbl.last_gather_kernel == 'k_ext_linterp'
    np.testing.assert_allclose(first.cpu().numpy(), direct.cpu().numpy(), rtol=1e-12)
    model.lbl.set_gather_mode('dynamic')
    model.run()
    with pytest.raises(RuntimeError, match='cannot be captured'):
        model.capture()                                    # (the direct gather can be: not tried here)
    profile = vt.flat()
    ec = model.ec.cpu().numpy()[:, 0]
    for layer in range(atm['nlayers']):
        want = np.zeros((1, g['nwave']))
        orc.extinction(want, profile, vt.size, vt.index, vg['lorentz'], vg['doppler'], g['wn'],
                       g['own'], g['divisors'], atm['dens'][layer], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoz'][:, layer].copy(), iso['isoiext'], ln['lwn'], ln['elow'],
                       ln['gf'], ln['lid'], vg['cutoff'], case['ethresh'], atm['temp'][layer],
                       0, 1, 1)
        assert np.array_equal(ec[layer] == 0, want[0] == 0)
        np.testing.assert_allclose(ec[layer], want[0], rtol=RTOL)
    assert np.count_nonzero(ec) > 0.5 * ec.size
    with pytest.raises(Exception, match='reference layout only'):
        eng.LBL(vt, model.lines, g['wn'][:10] * 0 + g['wn'][0] + 0.05 * np.arange(10),
                g['divisors'], atm['mol_radius'], atm['mol_mass'], iso['isoimol'],
                iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'], 1e-30)


def test_spectrum_without_materialised_depth(eng, case, orc, oracle_ec):
    """LBLSpectrum(materialize_depth=False): optical depths, the reference's exit rule and the
    transmission integral in one pass on the matrix cores, depth[L, W] / ideep[W] never written.
    The spectrum equals the default path's to 1e-13 and the oracle's to 1e-10; the timestamps keep
    the reference's three keys; two pipelines of one process share their side streams."""
    import torch
    atm = case['atm']
    full = eng.LBLSpectrum(case, rt_path='transit')
    want = full.run().clone()
    lean = eng.LBLSpectrum(case, rt_path='transit', voigt=full.voigt, lines=full.lines,
                           materialize_depth=False)
    got = lean.run()
    assert lean.depth is None and lean.ideep is None
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-13)
    depth, ideep = orc.optical_depth_transit(oracle_ec, atm['radius'], 0, atm['nlayers'],
                                             case['maxdepth'])
    ref = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL)
    assert list(lean.timestamps) == ['extinction', 'odepth', 'spectrum']
    p1 = eng.SpectrumPipeline(case, depth=2, voigt=full.voigt, lines=full.lines)
    p2 = eng.SpectrumPipeline(case, depth=2, voigt=full.voigt, lines=full.lines,
                              materialize_depth=False)
    assert all(a is b for a, b in zip(p1.streams, p2.streams))
    out, ev = p2.submit()
    ev.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=1e-13)
