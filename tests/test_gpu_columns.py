"""HIP column kernels (through the C ABI) against the golden vectors of the compiled
reference and against the oracle on seeded inputs.  Needs an MI355X.

Tolerance: rtol 1e-12 (binary64 throughout; device exp/pow differ from glibc by <= 1-2 ulp
and the sums run in the same order)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope='module')
def eng():
    import torch
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_transit_depth_and_transmission_golden(eng, golden, tag):
    g4, g5 = golden('g4_depth'), golden('g5_rt')
    itop, ibottom, maxdepth = g4[f'args_{tag}']
    itop, ibottom = int(itop), int(ibottom)
    ec = eng.dev(g4['ec'])
    path = eng.dev(eng.pack_raypath(eng.transit_path(g4['radius'], itop), itop))
    depth, ideep = eng.optical_depth_transit(ec, path, itop, ibottom, maxdepth)
    assert np.array_equal(host(ideep), g4[f'transit_ideep_{tag}'])
    np.testing.assert_allclose(host(depth), g4[f'transit_depth_{tag}'], rtol=RTOL)
    spec = eng.transmission(depth, ideep, eng.dev(g4['radius']), itop, float(g5['rstar']))
    np.testing.assert_allclose(host(spec), g5[f'transmission_{tag}'], rtol=RTOL)
    # fused entry: same depth / ideep / spectrum from one call
    spec2, depth2, ideep2 = eng.transit_spectrum(ec, path, eng.dev(g4['radius']),
                                                 float(g5['rstar']), itop, ibottom, maxdepth)
    assert np.array_equal(host(ideep2), host(ideep))
    assert np.array_equal(host(depth2), host(depth))
    assert np.array_equal(host(spec2), host(spec))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_plane_parallel_and_emission_golden(eng, golden, tag):
    import torch
    g4, g5 = golden('g4_depth'), golden('g5_rt')
    itop, ibottom, maxdepth = g4[f'args_{tag}']
    itop, ibottom = int(itop), int(ibottom)
    ec = eng.dev(g4['ec'])
    h = eng.dev(-np.diff(g4['radius']))
    depth, ideep = eng.plane_parallel_optical_depth(ec, h, itop, ibottom, maxdepth)
    assert np.array_equal(host(ideep), g4[f'plane_ideep_{tag}'])
    np.testing.assert_allclose(host(depth), g4[f'plane_depth_{tag}'], rtol=RTOL)
    # unfused: Planck table + intensity
    wn, temp, mu = eng.dev(g5['wn']), eng.dev(g5['temp']), eng.dev(g5['mu'])
    B = eng.blackbody_wn_2D(wn, temp)
    np.testing.assert_allclose(host(B), g5['B_full'], rtol=RTOL)
    ideep_i = eng.dev(g5[f'intensity_ideep_{tag}'], torch.int32)
    inten = eng.intensity(eng.dev(g4[f'plane_depth_{tag}']), ideep_i, B, mu, itop)
    np.testing.assert_allclose(host(inten), g5[f'intensity_{tag}'], rtol=1e-11, atol=1e-300)
    # fused: Planck in registers + quadrature
    weights = np.array([0.3, 0.25, 0.2, 0.15, 0.1])
    flux, inten2 = eng.emission_flux(eng.dev(g4[f'plane_depth_{tag}']), ideep_i, wn, temp, mu,
                                     eng.dev(weights), itop, want_intensity=True)
    np.testing.assert_allclose(host(inten2), g5[f'intensity_{tag}'], rtol=1e-11, atol=1e-300)
    want = np.sum(g5[f'intensity_{tag}'] * weights[:, None], axis=0)
    np.testing.assert_allclose(host(flux), want, rtol=1e-11)


def test_blackbody_last_and_1d(eng, golden):
    import ctypes as C
    import torch
    from pyratbay_amd._capi import call
    g5 = golden('g5_rt')
    wn, temp = eng.dev(g5['wn']), eng.dev(g5['temp'])
    B = eng.blackbody_wn_2D(wn, temp, eng.dev(g5['last'], torch.int32))
    np.testing.assert_allclose(host(B), g5['B_last'], rtol=RTOL)
    b1 = torch.empty_like(wn)
    call('pb_blackbody_wn', eng._ptr(b1), eng._ptr(wn), wn.shape[0], 1234.5, eng._stream())
    np.testing.assert_allclose(host(b1), g5['B_1d'], rtol=RTOL)


def test_single_optdepth_trapz_simps_ediff(eng, golden, orc):
    """The one-call-per-reference-function entry points against the oracle."""
    import torch
    from pyratbay_amd._capi import call
    c = cases.column_case(seed=9, nlayers=17, nwave=333)
    L, W = c['nlayers'], c['nwave']
    ec = eng.dev(c['ec'])
    rng = np.random.default_rng(1)
    # pb_optdepth on a row-slice view (data = ec[itop:r+1])
    itop, r = 2, 11
    path = orc.transit_path(c['radius'], itop)[r]
    ideep_h = np.where(rng.uniform(size=W) < 0.3, 5, -1).astype(np.int32)
    want_ideep = ideep_h.copy()
    want = orc.optdepth(c['ec'][itop:r + 1], path, 3.0, want_ideep, r)
    ideep = eng.dev(ideep_h, torch.int32)
    tau = torch.empty(W, dtype=torch.float64, device='cuda')
    # NB: every device buffer whose raw pointer is passed must stay referenced until the
    # call is issued (a temporary would be recycled by the caching allocator)
    path_d = eng.dev(path)
    call('pb_optdepth', eng._ptr(tau), ec[itop:].data_ptr(), W, eng._ptr(path_d),
         len(path), 3.0, eng._ptr(ideep), r, W, eng._stream())
    np.testing.assert_allclose(host(tau), want, rtol=RTOL)
    assert np.array_equal(host(ideep), want_ideep)
    # trapezoid2D with ragged nint
    nint_h = rng.integers(0, L, W).astype(np.int32)
    hh = rng.uniform(0.5, 2.0, L - 1)
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    hh_d, nint_d = eng.dev(hh), eng.dev(nint_h, torch.int32)
    call('pb_trapezoid2D', eng._ptr(out), eng._ptr(ec), eng._ptr(hh_d), eng._ptr(nint_d),
         L, W, eng._stream())
    np.testing.assert_allclose(host(out), orc.trapezoid2D(c['ec'], hh, nint_h), rtol=RTOL)
    # simps2D (odd and even row counts)
    for ny in (L, L - 1):
        y = c['ec'][:ny] + 1e-12
        hs, hr, hf = orc.geth(hh[:ny - 1])
        nint_s = rng.integers(0, ny + 1, W).astype(np.int32)
        bufs = [eng.dev(y), eng.dev(hh[:ny - 1]), eng.dev(nint_s, torch.int32),
                eng.dev(hs), eng.dev(hr), eng.dev(hf)]
        call('pb_simps2D', eng._ptr(out), eng._ptr(bufs[0]), ny, W, eng._ptr(bufs[1]),
             eng._ptr(bufs[2]), eng._ptr(bufs[3]), eng._ptr(bufs[4]), eng._ptr(bufs[5]),
             eng._stream())
        np.testing.assert_allclose(host(out), orc.simps2D(y, hh[:ny - 1], nint_s, hs, hr, hf),
                                   rtol=RTOL)
    # ediff
    d = torch.empty(L - 1, dtype=torch.float64, device='cuda')
    rad_d = eng.dev(c['radius'])
    call('pb_ediff', eng._ptr(d), eng._ptr(rad_d), L, eng._stream())
    assert np.array_equal(host(d), orc.ediff(c['radius']))


@pytest.mark.parametrize('shape', [(1, 1), (2, 63), (5, 64), (80, 1000), (120, 257)])
def test_transit_vs_oracle_shapes(eng, orc, shape):
    """Ragged sizes incl. single layer / single column; itop>0; every column thick or thin."""
    L, W = shape
    c = cases.column_case(seed=L * 1000 + W, nlayers=L, nwave=W)
    for itop, ibottom, maxdepth in ((0, L, 10.0), (min(1, L - 1), L, 0.5), (0, max(L - 1, 0), np.inf)):
        want_d, want_i = orc.optical_depth_transit(c['ec'], c['radius'], itop, ibottom, maxdepth)
        path = eng.dev(eng.pack_raypath(eng.transit_path(c['radius'], itop), itop))
        depth, ideep = eng.optical_depth_transit(eng.dev(c['ec']), path, itop, ibottom, maxdepth)
        assert np.array_equal(host(ideep), want_i)
        np.testing.assert_allclose(host(depth), want_d, rtol=RTOL)
        spec = eng.transmission(depth, ideep, eng.dev(c['radius']), itop, c['rstar'])
        np.testing.assert_allclose(
            host(spec), orc.transmission(want_d, c['radius'], c['rstar'], want_i, itop),
            rtol=RTOL)


def test_clear_atmosphere_kats(eng):
    """Analytic KATs of the reference's tests (test_transmission.py:43-52,
    test_emission.py:43-52): zero extinction -> (r_bottom/R*)**2 and B(T_bottom)."""
    import torch
    c = cases.column_case()
    L, W = c['nlayers'], c['nwave']
    ec = torch.zeros((L, W), dtype=torch.float64, device='cuda')
    path = eng.dev(eng.pack_raypath(eng.transit_path(c['radius'], 0), 0))
    depth, ideep = eng.optical_depth_transit(ec, path, 0, L, 10.0)
    spec = eng.transmission(depth, ideep, eng.dev(c['radius']), 0, c['rstar'])
    np.testing.assert_allclose(host(spec), (c['radius'][-1] / c['rstar'])**2, rtol=1e-14)
    pd, pi = eng.plane_parallel_optical_depth(ec, eng.dev(-np.diff(c['radius'])), 0, L, 10.0)
    flux, inten = eng.emission_flux(pd, pi, eng.dev(c['wn']), eng.dev(c['temp']),
                                    eng.dev(c['mu']), eng.dev(np.ones(5)), 0, True)
    B = host(eng.blackbody_wn_2D(eng.dev(c['wn']), eng.dev(c['temp'])))
    for k in range(5):
        np.testing.assert_allclose(host(inten)[k], B[-1], rtol=1e-13)


def test_interp_ec_golden(eng, golden):
    import torch
    g = golden('g3_interp')
    nmol, ntemp, nlayers, nwave = g['etable'].shape
    et, tt = eng.dev(g['etable']), eng.dev(g['ttable'])
    te, de = eng.dev(g['temps']), eng.dev(g['dens'])
    a = torch.full((nlayers, nwave), 1e-12, dtype=torch.float64, device='cuda')
    eng.interp_ec(a, et, tt, te, de, 0, nlayers)
    np.testing.assert_allclose(host(a), g['full'], rtol=RTOL)
    b = torch.zeros((nlayers, nwave), dtype=torch.float64, device='cuda')
    eng.interp_ec(b, et, tt, te, de, 2, 5)
    np.testing.assert_allclose(host(b), g['part'], rtol=RTOL)
    m = torch.zeros((nmol, nlayers, nwave), dtype=torch.float64, device='cuda')
    eng.interp_ec(m, et, tt, te, de, 0, nlayers + 3, per_mol=True)
    np.testing.assert_allclose(host(m), g['per_mol'], rtol=RTOL)
    # the assigning form: same values as accumulating into zeros, whatever the rows held,
    # and rows outside [lay1, lay2) untouched
    junk = torch.full((nlayers, nwave), 7.5, dtype=torch.float64, device='cuda')
    eng.interp_ec(junk, et, tt, te, de, 2, 5, assign=True)
    want = g['part'].copy()
    want[:2] = 7.5
    want[5:] = 7.5
    assert np.array_equal(host(junk)[:2], want[:2]) and np.array_equal(host(junk)[5:], want[5:])
    assert np.array_equal(host(junk)[2:5], host(b)[2:5])
    mj = torch.full((nmol, nlayers, nwave), -3.0, dtype=torch.float64, device='cuda')
    eng.interp_ec(mj, et, tt, te, de, 0, nlayers, per_mol=True, assign=True)
    assert np.array_equal(host(mj), host(m))


def test_loglike_reference_fixture(eng, golden):
    """pb_loglike against fixture G14 = the reference's own Loglike.__call__
    (tools/retrieval_tools.py:73-104) on nine band-integrated models: rejected models (inf or
    nan band flux) give the reference's -1e98, a perfect fit the normalisation term alone."""
    g = golden('g14_loglike')
    got = eng.loglike(eng.dev(g['models']), eng.dev(g['data']), eng.dev(g['uncert'])).cpu().numpy()
    assert got[3] == -1.0e98 and got[6] == -1.0e98
    np.testing.assert_allclose(got, g['loglike'], rtol=1e-13)
    one = eng.loglike(eng.dev(g['models'][4]), eng.dev(g['data']), eng.dev(g['uncert']))
    np.testing.assert_allclose(one.cpu().numpy(), g['loglike'][4:5], rtol=1e-13)


@pytest.mark.parametrize('seed', range(8))
def test_random_rt_configurations(eng, orc, seed):
    """Randomly drawn column problems -- layers, columns (narrow grids take the 4-rows-per-
    thread transit kernels, wide ones the 16-row form), top layer, maximum depth, with and
    without an opaque cloud deck -- through the fused and the two-call transit forms and the
    plane-parallel emission path, against the oracle."""
    rng = np.random.default_rng(500 + seed)
    L = int(rng.integers(2, 90))
    W = int(rng.choice([1, 63, 700, 40000]))
    c = cases.column_case(seed=seed + 77, nlayers=L, nwave=W)
    itop = int(rng.integers(0, max(1, L // 3)))
    maxdepth = float(rng.choice([0.3, 10.0, np.inf]))
    deck_itop = int(rng.integers(itop + 1, L)) if rng.random() < 0.5 and L - itop > 2 else None
    ibottom = L if deck_itop is None else deck_itop + 1
    radius, ec = c['radius'], c['ec']
    rsurf = tsurf = None
    if deck_itop is not None:
        f = float(rng.uniform(0.05, 0.95))          # cloud top between two layers
        rsurf = radius[deck_itop - 1] + f * (radius[deck_itop] - radius[deck_itop - 1])
        tsurf = c['temp'][deck_itop - 1] + f * (c['temp'][deck_itop] - c['temp'][deck_itop - 1])
    # transit
    want_d, want_i = orc.optical_depth_transit(ec, radius, itop, ibottom, maxdepth)
    want = orc.transmission_deck(want_d, radius, c['rstar'], want_i, itop, rsurf, deck_itop)
    path = eng.dev(eng.pack_raypath(eng.transit_path(radius, itop), itop))
    spec, depth, ideep = eng.transit_spectrum(eng.dev(ec), path, eng.dev(radius), c['rstar'],
                                              itop, ibottom, maxdepth, rsurf, deck_itop)
    assert np.array_equal(host(ideep), want_i)
    np.testing.assert_allclose(host(depth), want_d, rtol=RTOL)
    np.testing.assert_allclose(host(spec), want, rtol=RTOL)
    depth2, ideep2 = eng.optical_depth_transit(eng.dev(ec), path, itop, ibottom, maxdepth)
    spec2 = eng.transmission(depth2, ideep2, eng.dev(radius), itop, c['rstar'], rsurf, deck_itop)
    assert np.array_equal(host(spec2), host(spec))
    # plane-parallel emission
    h = -orc.ediff(radius)
    wd, wi = np.zeros((L, W)), np.full(W, L - 1, np.int32)
    orc.plane_parallel_optical_depth(wd, wi, ec, h, maxdepth, itop, ibottom)
    weights = np.pi * np.diff(np.sin(np.radians([0, 10, 30, 50, 70, 90]))**2)
    want_f = orc.emission_deck(wd, wi, c['wn'], c['temp'], c['mu'], weights, itop, tsurf,
                               deck_itop)
    d3, i3 = eng.plane_parallel_optical_depth(eng.dev(ec), eng.dev(h), itop, ibottom, maxdepth)
    assert np.array_equal(host(i3), wi)
    np.testing.assert_allclose(host(d3), wd, rtol=RTOL)
    flux = eng.emission_flux(d3, i3, eng.dev(c['wn']), eng.dev(c['temp']), eng.dev(c['mu']),
                             eng.dev(weights), itop, cloud_tsurf=tsurf, cloud_itop=deck_itop)
    np.testing.assert_allclose(host(flux), want_f, rtol=RTOL)


@pytest.mark.parametrize('nlayers', [1, 2, 9, 17, 40])
def test_plane_parallel_depth_stop_rules(eng, orc, nlayers):
    """k_plane_depth fetches its rows eight at a time before it applies the reference's in-order
    stop (maxdepth reached, ibottom, last layer): every top layer (the last one included: no row
    is examined), every bottom, stops in the first / a middle / the last row of a fetch group and
    no stop at all, against the oracle -- depth bit for bit (same sums), ideep exact."""
    rng = np.random.default_rng(nlayers)
    W = 777
    ec = 10.0**rng.uniform(-11, -8, (nlayers, W))
    radius = np.linspace(7.5e9, 7.0e9, max(nlayers, 2))[:nlayers]
    h = -np.diff(radius) if nlayers > 1 else np.zeros(0)
    col_total = 0.5 * np.sum(h[:, None] * (ec[1:] + ec[:-1]), axis=0) if nlayers > 1 else np.zeros(W)
    ec_d, h_d = eng.dev(ec), eng.dev(h if nlayers > 1 else np.zeros(1))
    tops = sorted({0, 1, nlayers // 2, nlayers - 2, nlayers - 1} & set(range(nlayers)))
    for itop in tops:
        for ibottom in sorted({itop, itop + 1, nlayers // 2, nlayers - 1, nlayers} & set(range(nlayers + 1))):
            for maxdepth in (0.0, float(np.median(col_total)) * 0.03, float(np.median(col_total)) * 0.5,
                             np.inf):
                wd, wi = np.zeros((nlayers, W)), np.full(W, nlayers - 1, np.int32)
                orc.plane_parallel_optical_depth(wd, wi, ec, h if nlayers > 1 else np.zeros(1),
                                                 maxdepth, itop, ibottom)
                d, i = eng.plane_parallel_optical_depth(ec_d, h_d, itop, ibottom, maxdepth)
                what = (nlayers, itop, ibottom, maxdepth)
                assert np.array_equal(host(i), wi), what
                assert np.array_equal(host(d), wd), what


def test_special_values_in_the_short_divisions(eng, orc):
    """pb::quot (x / d as product + exact residual + one correction) must behave like the
    reference's C division for the special values it can meet: an infinite
    optical depth gives exp(-inf / mu) = 0, not NaN; a layer at T = 0 radiates B = 0, not NaN.
    The intensity of columns whose deepest layers are opaque to infinity, and the Planck function
    of a zero-temperature layer, against the oracle (IEEE division, no -ffast-math)."""
    import torch
    c = cases.column_case(seed=17, nlayers=12, nwave=70)
    L, W = c['nlayers'], c['nwave']
    depth = np.cumsum(np.abs(c['ec']) * 1e9, axis=0)
    depth[0] = 0.0
    depth[8:, ::3] = np.inf                       # opaque to infinity below layer 8
    depth[5:, 1::7] = np.inf
    ideep = np.full(W, L - 1, np.int32)
    temp = c['temp'].copy()
    temp[3] = 0.0
    B = host(eng.blackbody_wn_2D(eng.dev(c['wn']), eng.dev(temp)))
    want_B = orc.blackbody_wn_2D(c['wn'], temp)
    assert np.all(np.isfinite(B)) and np.all(B[3] == 0.0) and np.all(want_B[3] == 0.0)
    np.testing.assert_allclose(B, want_B, rtol=RTOL)
    want = orc.intensity(depth, ideep, want_B, c['mu'], 0)
    assert np.all(np.isfinite(want))
    got = host(eng.intensity(eng.dev(depth), eng.dev(ideep, torch.int32), eng.dev(want_B),
                             eng.dev(c['mu']), 0))
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-300)
    weights = np.array([0.3, 0.25, 0.2, 0.15, 0.1])
    flux = host(eng.emission_flux(eng.dev(depth), eng.dev(ideep, torch.int32), eng.dev(c['wn']),
                                  eng.dev(temp), eng.dev(c['mu']), eng.dev(weights), 0))
    np.testing.assert_allclose(flux, np.sum(want * weights[:, None], axis=0), rtol=1e-11)


@pytest.mark.parametrize('n', [1, 5, 16])
def test_emission_flux_gauss_quadrature(eng, orc, n):
    """`quadrature = n` (pyrat/spectrum.py:41-49, 366-377): the emission flux over the
    Gauss-Legendre angles of engine.gauss_quadrature against the oracle chain (plane-parallel
    depth -> Planck -> intensity per mu -> weighted sum); n = 16 fills every running sum of the
    kernel."""
    c = cases.column_case(seed=23 + n, nlayers=30, nwave=333)
    mu, weights = eng.gauss_quadrature(n)
    h = -np.diff(c['radius'])
    depth, ideep = eng.plane_parallel_optical_depth(eng.dev(c['ec']), eng.dev(h), 0,
                                                    c['nlayers'], 10.0)
    flux, inten = eng.emission_flux(depth, ideep, eng.dev(c['wn']), eng.dev(c['temp']),
                                    eng.dev(mu), eng.dev(weights), 0, want_intensity=True)
    wd = np.zeros_like(c['ec'])
    wi = np.zeros(c['nwave'], np.int32)
    orc.plane_parallel_optical_depth(wd, wi, c['ec'], h, 10.0, 0, c['nlayers'])
    B = orc.blackbody_wn_2D(c['wn'], c['temp'])
    want = orc.intensity(wd, wi, B, mu, 0)
    np.testing.assert_allclose(host(inten), want, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(host(flux), np.sum(want * weights[:, None], axis=0), rtol=1e-11)
