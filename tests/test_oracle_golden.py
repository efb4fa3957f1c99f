"""The oracle (oracle/pb_oracle.c) against golden vectors produced by the compiled,
unmodified reference C (tests/golden/make_golden.py).  CPU only.

Tolerance: 1e-13 relative -- the restatement follows the reference's operation
order; the residual is -ffast-math re-association in the reference build."""
import numpy as np
import pytest

import cases

RTOL = 1e-13


def close(a, b, rtol=RTOL, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_g1_voigt_table(orc, golden):
    g = golden('g1_voigt')
    size = g['size_in'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, g['lorentz'], g['doppler'], float(g['dwn']))
    assert np.array_equal(size, g['size_out'])
    assert np.array_equal(index, g['index_out'])
    q0, qn, st = int(g['quick_start']), int(g['quick_n']), int(g['quick_stride'])
    close(profile[:q0], g['profile_head'])
    close(profile[q0:q0 + qn:st], g['quick_sub'])
    close(np.sum(profile[q0:q0 + qn]), g['quick_sum'], rtol=1e-12)
    # the tail of the buffer (aliased cells were allocated 1 sample each) is untouched
    assert np.all(profile[int(g['used']):] == 0)


@pytest.mark.parametrize('mode', ['step', 'res'])
def test_g2_extinction(orc, golden, mode):
    g = golden('g2_extinction')
    c = cases.extinction_inputs(resolution=(mode == 'res'))
    size = c['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, c['lorentz'], c['doppler'],
                   c['own'][1] - c['own'][0])
    assert np.array_equal(size, g[f'size_out_{mode}'])
    assert np.array_equal(index, g[f'index_out_{mode}'])
    atm, iso = c['atm'], c['iso']
    for k, (layer, add, cut, eth, skip) in enumerate(cases.extinction_variants()):
        isoiext = iso['isoiext'].copy()
        if skip:
            isoiext[1] = -1
        temp = atm['temp'][layer]
        ext = np.zeros((1 if add else c['nspec'], len(c['wn'])))
        st = orc.extinction(
            ext, profile, size, index, c['lorentz'], c['doppler'], c['wn'], c['own'],
            c['divisors'], atm['dens'][layer], atm['mol_radius'], atm['mol_mass'],
            iso['isoimol'], iso['isomass'], iso['isoratio'], cases.iso_z(temp, 3),
            isoiext, c['lwn'], c['elow'], c['gf'], c['lid'],
            c['cutoff'] if cut else 0.0, eth, temp, 0, add, int(mode == 'res'),
            return_stats=True)
        want = g[f'ext_{mode}'][k][:ext.shape[0]]
        assert np.array_equal(ext == 0, want == 0), (k, st)
        close(ext, want, rtol=1e-12)
        assert st['nadd'] > 0
        if eth > 1e-10:
            assert st['nskip'] > 0


def test_g3_interp(orc, golden):
    g = golden('g3_interp')
    nmol, ntemp, nlayers, nwave = g['etable'].shape
    a = np.full((nlayers, nwave), 1e-12)
    orc.interp_ec(a, g['etable'], g['ttable'], g['temps'], g['dens'], 0, nlayers)
    close(a, g['full'])
    b = np.zeros((nlayers, nwave))
    orc.interp_ec(b, g['etable'], g['ttable'], g['temps'], g['dens'], 2, 5)
    close(b, g['part'])
    assert np.all(b[:2] == 0) and np.all(b[5:] == 0)
    m = np.zeros((nmol, nlayers, nwave))
    orc.interp_ec_per_mol(m, g['etable'], g['ttable'], g['temps'], g['dens'], 0,
                          nlayers + 3)
    close(m, g['per_mol'])


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_g4_depth(orc, golden, tag):
    g = golden('g4_depth')
    itop, ibottom, maxdepth = g[f'args_{tag}']
    itop, ibottom = int(itop), int(ibottom)
    L, W = g['ec'].shape
    depth, ideep = orc.optical_depth_transit(g['ec'], g['radius'], itop, ibottom,
                                             maxdepth)
    assert np.array_equal(ideep, g[f'transit_ideep_{tag}'])
    close(depth, g[f'transit_depth_{tag}'])
    pd = np.zeros((L, W))
    pi = np.full(W, L - 1, np.int32)
    orc.plane_parallel_optical_depth(pd, pi, g['ec'], -orc.ediff(g['radius']), maxdepth,
                                     itop, ibottom)
    assert np.array_equal(pi, g[f'plane_ideep_{tag}'])
    close(pd, g[f'plane_depth_{tag}'])


def test_g5_rt(orc, golden):
    g = golden('g5_rt')
    g4 = golden('g4_depth')
    close(orc.blackbody_wn_2D(g['wn'], g['temp']), g['B_full'])
    B = np.zeros_like(g['B_last'])
    orc.blackbody_wn_2D(g['wn'], g['temp'], B, g['last'])
    close(B, g['B_last'])
    close(orc.blackbody_wn(g['wn'], 1234.5), g['B_1d'])
    for tag in 'abc':
        itop = int(g4[f'args_{tag}'][0])
        out = orc.intensity(g4[f'plane_depth_{tag}'], g[f'intensity_ideep_{tag}'],
                            g['B_full'], g['mu'], itop)
        close(out, g[f'intensity_{tag}'], rtol=1e-12, atol=1e-300)
        spec = orc.transmission(g4[f'transit_depth_{tag}'], g['radius'],
                                float(g['rstar']), g4[f'transit_ideep_{tag}'], itop)
        close(spec, g[f'transmission_{tag}'])
    close(orc.trapezoid(g['trap1d_y'], g['trap1d_h']), g['trap1d'])
    out = np.zeros(12)
    assert orc.cumulative_sum(out, g['trap1d_y'], g['trap1d_h'], 0.9) == int(g['cumsum_n'])
    close(out, g['cumsum_out'])
    close(orc.ediff(g['radius']), g['ediff'])
    for tag in ('odd', 'even'):
        x, y = g[f'simps_x_{tag}'], g[f'simps_y_{tag}']
        h = np.diff(x)
        hs, hr, hf = orc.geth(h)
        close(hs, g[f'simps_hsum_{tag}'])
        close(hr, g[f'simps_hratio_{tag}'])
        close(hf, g[f'simps_hfactor_{tag}'])
        close(orc.simps(y[:, 0].copy(), h, hs, hr, hf), g[f'simps_1d_{tag}'])
        close(orc.simps2D(y, h, g[f'simps_nint_{tag}'], hs, hr, hf), g[f'simps_2d_{tag}'])
    assert orc.geth(np.empty(0)) == [0, 0, 0]
    assert [orc.ifirst(g['flags']), orc.ifirst(np.zeros(4, np.int32), -7)] == list(g['ifirst'])
    assert [orc.ilast(g['flags']), orc.ilast(np.zeros(4, np.int32))] == list(g['ilast'])


def test_analytic_kats(orc):
    """KATs that need no line data, restated from the reference's tests:
    clear transmission == (r_bottom/R*)**2 (tests/test_transmission.py:43-52),
    clear emission == pi*B(T_bottom) for an isotropic quadrature
    (tests/test_emission.py:43-52), simps == exact for a parabola
    (tests/test_src.py:42-54)."""
    c = cases.column_case()
    L, W = c['nlayers'], c['nwave']
    ec = np.zeros((L, W))
    depth, ideep = orc.optical_depth_transit(ec, c['radius'], 0, L, 10.0)
    assert np.all(depth == 0) and np.all(ideep == L - 1)
    spec = orc.transmission(depth, c['radius'], c['rstar'], ideep, 0)
    close(spec, np.full(W, (c['radius'][-1] / c['rstar'])**2), rtol=1e-14)
    pd = np.zeros((L, W))
    pi = np.full(W, L - 1, np.int32)
    orc.plane_parallel_optical_depth(pd, pi, ec, -orc.ediff(c['radius']), 10.0, 0, L)
    B = orc.blackbody_wn_2D(c['wn'], c['temp'])
    inten = orc.intensity(pd, pi, B, c['mu'], 0)
    for k in range(len(c['mu'])):
        close(inten[k], B[-1], rtol=1e-14)
    x = np.linspace(0, 2, 9)
    h = np.diff(x)
    hs, hr, hf = orc.geth(h)
    close(orc.simps(3 * x**2, h, hs, hr, hf), 8.0, rtol=1e-14)
