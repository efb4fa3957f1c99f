"""The drop-in modules pyratbay_amd.lib.* called exactly like the reference's extension
modules (positional signatures, int64 index arrays, in-place outputs), checked against
the golden vectors of the compiled reference and mirroring the reference's own native
unit tests (tests/test_src.py:42-98).  Needs an MI355X."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope='module')
def hip():
    from pyratbay_amd import engine
    engine.require_gpu()
    import pyratbay_amd.lib as lib
    return lib


def test_module_surface(hip):
    """Same public names as the reference's PyMethodDef tables."""
    want = {
        '_extcoeff': ['extinction', 'interp_ec', 'interp_ec_per_mol'],
        'vprofile': ['grid'],
        '_trapezoid': ['trapezoid', 'trapezoid2D', 'cumulative_sum',
                       'plane_parallel_optical_depth', 'optdepth', 'intensity'],
        '_simpson': ['geth', 'simps', 'simps2D'],
        '_blackbody': ['blackbody_wn_2D', 'blackbody_wn'],
        'cutils': ['ediff', 'arrbinsearch'],
        '_indices': ['ifirst', 'ilast'],
    }
    for mod, names in want.items():
        for name in names:
            assert callable(getattr(getattr(hip, mod), name)), f'{mod}.{name}'


def test_vprofile_grid_in_place(hip, golden):
    g = golden('g1_voigt')
    size = g['size_in'].astype(int)            # int64, like voigt.py:109
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1), np.double)
    assert hip.vprofile.grid(profile, size, index, g['lorentz'], g['doppler'],
                             float(g['dwn']), 0) == 1
    assert np.array_equal(size, g['size_out']) and np.array_equal(index, g['index_out'])
    q0 = int(g['quick_start'])
    np.testing.assert_allclose(profile[:q0], g['profile_head'], rtol=2e-12)
    assert np.all(profile[int(g['used']):] == 0)


@pytest.mark.parametrize('mode', ['step', 'res'])
def test_extinction_per_layer_calls(hip, golden, orc, mode):
    """The loop of pyratbay/pyrat/extinction.py:170-213: one call per layer, ext zeroed by
    the caller, int64 index arrays, non-contiguous isoz column."""
    g = golden('g2_extinction')
    c = cases.extinction_inputs(resolution=(mode == 'res'))
    size = c['size'].astype(int)
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    hip.vprofile.grid(profile, size, index, c['lorentz'], c['doppler'],
                      c['own'][1] - c['own'][0], 0)
    atm, iso = c['atm'], c['iso']
    iso_pf = np.array([cases.iso_z(t, 3) for t in atm['temp']]).T        # [niso, nlayers]
    for k, (layer, add, cut, eth, skip) in enumerate(cases.extinction_variants()):
        isoiext = iso['isoiext'].astype(int)
        if skip:
            isoiext[1] = -1
        ext = np.zeros((1 if add else c['nspec'], len(c['wn'])))
        ret = hip._extcoeff.extinction(
            ext, profile, size, index, c['lorentz'], c['doppler'], c['wn'], c['own'],
            c['divisors'].astype(int), atm['dens'][layer], atm['mol_radius'],
            atm['mol_mass'], iso['isoimol'].astype(int), iso['isomass'], iso['isoratio'],
            iso_pf[:, layer], isoiext, c['lwn'], c['elow'], c['gf'], c['lid'].astype(int),
            c['cutoff'] if cut else 0.0, eth, atm['temp'][layer], -8, add,
            int(mode == 'res'))
        assert ret == 1
        want = g[f'ext_{mode}'][k][:ext.shape[0]]
        assert np.array_equal(ext == 0, want == 0), k
        np.testing.assert_allclose(ext, want, rtol=RTOL, err_msg=f'variant {k}')


def test_interp_ec(hip, golden):
    g = golden('g3_interp')
    nmol, ntemp, nlayers, nwave = g['etable'].shape
    a = np.full((nlayers, nwave), 1e-12)
    assert hip._extcoeff.interp_ec(a, g['etable'], g['ttable'], g['temps'], g['dens'], 0,
                                   nlayers) == 1
    np.testing.assert_allclose(a, g['full'], rtol=1e-12)
    m = np.zeros((nmol, nlayers, nwave))
    hip._extcoeff.interp_ec_per_mol(m, g['etable'], g['ttable'], g['temps'], g['dens'], 0,
                                    nlayers + 3)
    np.testing.assert_allclose(m, g['per_mol'], rtol=1e-12)


def test_optical_depth_loop_like_optic_depth_py(hip, golden):
    """pyratbay/opacity/optic_depth.py:103-130 with the drop-in _trapezoid / cutils."""
    from pyratbay_amd.engine import transit_path
    g4, g5 = golden('g4_depth'), golden('g5_rt')
    t, cu = hip._trapezoid, hip.cutils
    for tag in 'abc':
        itop, ibottom, maxdepth = g4[f'args_{tag}']
        itop, ibottom = int(itop), int(ibottom)
        ec, radius = g4['ec'], g4['radius']
        nlayers, nwave = ec.shape
        raypath = transit_path(radius, itop)
        depth = np.zeros((nlayers, nwave))
        ideep = np.array(np.tile(-1, nwave), dtype=np.intc)
        r = itop
        for r in range(itop, ibottom):
            depth[r] = t.optdepth(ec[itop:r + 1], raypath[r], maxdepth, ideep, r)
        ideep[ideep < 0] = r
        assert np.array_equal(ideep, g4[f'transit_ideep_{tag}'])
        np.testing.assert_allclose(depth, g4[f'transit_depth_{tag}'], rtol=1e-12)
        # transmission (radiative_transfer.py:57-71)
        h = np.ediff1d(radius[itop:])
        integ = np.exp(-depth[itop:]) * np.expand_dims(radius[itop:], 1)
        spectrum = t.trapezoid2D(integ, h, (ideep - itop + 1) - 1)
        spectrum = (radius[itop]**2 + 2 * spectrum) / float(g5['rstar'])**2
        np.testing.assert_allclose(spectrum, g5[f'transmission_{tag}'], rtol=1e-12)
        # emission branch
        pdepth = np.zeros((nlayers, nwave))
        pideep = np.tile(nlayers - 1, nwave)                       # int64
        assert t.plane_parallel_optical_depth(pdepth, pideep, ec, -cu.ediff(radius),
                                              maxdepth, itop, ibottom) is None
        assert np.array_equal(pideep, g4[f'plane_ideep_{tag}'])
        np.testing.assert_allclose(pdepth, g4[f'plane_depth_{tag}'], rtol=1e-12)
        B = hip._blackbody.blackbody_wn_2D(g5['wn'], g5['temp'])
        inten = t.intensity(g4[f'plane_depth_{tag}'], g5[f'intensity_ideep_{tag}'], B,
                            g5['mu'], itop)
        np.testing.assert_allclose(inten, g5[f'intensity_{tag}'], rtol=1e-11, atol=1e-300)


def test_blackbody_variants(hip, golden):
    g5 = golden('g5_rt')
    B = np.zeros_like(g5['B_last'])
    assert hip._blackbody.blackbody_wn_2D(g5['wn'], g5['temp'], B, g5['last']) == 1
    np.testing.assert_allclose(B, g5['B_last'], rtol=1e-12)
    np.testing.assert_allclose(hip._blackbody.blackbody_wn(g5['wn'], 1234.5), g5['B_1d'],
                               rtol=1e-12)
    b1 = np.zeros(len(g5['wn']))
    assert hip._blackbody.blackbody_wn(g5['wn'], 1234.5, b1) == 1
    np.testing.assert_allclose(b1, g5['B_1d'], rtol=1e-12)


def test_simps_like_test_src(hip, golden):
    """tests/test_src.py:42-54: geth + simps against scipy's Simpson rule, odd and even."""
    import scipy.integrate as si
    g5 = golden('g5_rt')
    x = np.linspace(0, 3.3, 21)
    y = np.sin(x)**2 + 1
    for n in (21, 20):
        h = np.ediff1d(x[:n])
        hsum, hratio, hfactor = hip._simpson.geth(h)
        got = hip._simpson.simps(y[:n], h, hsum, hratio, hfactor)
        # the reference pairs intervals from index len(h)%2 and uses the trapezoid rule on
        # the last interval for even sample counts; for a uniform grid that equals scipy's
        # composite rule on the first n-1 (odd) samples + trapezoid
        if n % 2:
            np.testing.assert_allclose(got, si.simpson(y[:n], x=x[:n]), rtol=1e-13)
    assert hip._simpson.geth(np.array([])) == [0, 0, 0]
    for tag in ('odd', 'even'):
        xx, yy = g5[f'simps_x_{tag}'], g5[f'simps_y_{tag}']
        h = np.diff(xx)
        hs, hr, hf = hip._simpson.geth(h)
        np.testing.assert_allclose(hs, g5[f'simps_hsum_{tag}'], rtol=1e-15)
        np.testing.assert_allclose(hip._simpson.simps(yy[:, 0], h, hs, hr, hf),
                                   g5[f'simps_1d_{tag}'], rtol=1e-13)
        np.testing.assert_allclose(
            hip._simpson.simps2D(yy, h, g5[f'simps_nint_{tag}'], hs, hr, hf),
            g5[f'simps_2d_{tag}'], rtol=1e-13)


def test_cutils_and_indices_like_test_src(hip, golden):
    """tests/test_src.py:57-98"""
    g5 = golden('g5_rt')
    np.testing.assert_array_equal(hip.cutils.ediff(g5['radius']), g5['ediff'])
    np.testing.assert_allclose(hip.cutils.ediff(np.array([1.0, 3.0, 7.5])), [2.0, 4.5])
    data = np.array([1, 1, 0, 0, 0], np.intc)
    assert hip._indices.ifirst(data) == 0 and hip._indices.ilast(data) == 1
    data = np.array([0, 0, 1, 1, 0, 1])
    assert hip._indices.ifirst(data) == 2 and hip._indices.ilast(data) == 5
    assert hip._indices.ifirst(np.zeros(5, int)) == -1
    assert hip._indices.ilast(np.zeros(5, int), 0) == 0
    np.testing.assert_allclose(hip._trapezoid.trapezoid(g5['trap1d_y'], g5['trap1d_h']),
                               g5['trap1d'], rtol=1e-13)
    out = np.zeros(12)
    assert hip._trapezoid.cumulative_sum(out, g5['trap1d_y'], g5['trap1d_h'], 0.9) == \
        int(g5['cumsum_n'])
    np.testing.assert_allclose(out, g5['cumsum_out'], rtol=1e-14)
    idx = hip.cutils.arrbinsearch(np.array([0.1, 2.4, 2.6, 9.0]), np.array([1., 2., 3., 4.]))
    assert list(idx) == [0, 1, 2, 3] and idx.dtype == np.int32


def _dropin_case():
    from pyratbay_amd import engine, synth
    case = synth.lbl_case(6001, 3, 30000, wnosamp=24, nlor=12, ndop=6, extent=60.0, cutoff=3.0,
                          niso=2, seed=21)
    g, vg = case['grid'], case['voigt']
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 24, True)
    profile, psize, pindex = vt.flat(), np.array(vt.size, int), np.array(vt.index, int)
    vt.close()
    return case, profile, psize, pindex


def _dropin_call(hip, case, profile, psize, pindex, layer, lines=None):
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    ln = lines or ln
    # the package holds its arrays for the whole run: pass the SAME int64 copy of the line ids every
    # time (a fresh astype() per call is a new buffer, which the cache rightly hashes in full)
    if 'lid64' not in ln:
        ln['lid64'] = ln['lid'].astype(int)
    ext = np.zeros((1, g['nwave']))
    hip._extcoeff.extinction(ext, profile, psize, pindex, vg['lorentz'], vg['doppler'], g['wn'],
                             g['own'], g['divisors'].astype(int), atm['dens'][layer],
                             atm['mol_radius'], atm['mol_mass'], iso['isoimol'].astype(int),
                             iso['isomass'], iso['isoratio'], iso['isoz'][:, layer],
                             iso['isoiext'].astype(int), ln['lwn'], ln['elow'], ln['gf'],
                             ln['lid64'], vg['cutoff'], case['ethresh'],
                             float(atm['temp'][layer]), 0, 1, 0)
    return ext


def test_extinction_cache_sees_in_place_edits(hip, monkeypatch):
    """The drop-in keeps the Voigt table and the line list on the device between the per-layer
    calls, keyed on the caller's buffers: identity + a 64-KiB probe per call, a full content
    hash only for a buffer not seen before or whose probe changed.  An in-place edit of a cached
    array (here: all line strengths scaled; the table scaled) must still be seen, an unchanged
    buffer must not be re-hashed, and a different array of equal content must hit the cache."""
    mod = hip._extcoeff
    mod.invalidate()
    # (the probe mechanics are exercised on this small case by lowering the size from which
    # buffers are probed instead of hashed in full; the default is 32 MiB)
    monkeypatch.setattr(mod, '_SMALL_BYTES', 64 << 10)
    case, profile, psize, pindex = _dropin_case()
    full = []
    real_digest = mod._digest
    monkeypatch.setattr(mod, '_digest', lambda a: (full.append(np.asarray(a).nbytes),
                                                   real_digest(a))[1])
    base = _dropin_call(hip, case, profile, psize, pindex, 1)
    first = len(full)
    assert first >= 5                                   # table, own, the line arrays
    again = _dropin_call(hip, case, profile, psize, pindex, 1)
    assert np.array_equal(again, base) and len(full) == first      # probes only
    voigt_before = mod._cache['voigt']
    # equal content at another address: full hash of the new buffer, cached device copy kept
    copy = {k: v.copy() for k, v in case['lines'].items() if k != 'lid64'}
    assert np.array_equal(_dropin_call(hip, case, profile, psize, pindex, 1, lines=copy), base)
    assert len(full) > first and mod._cache['voigt'] is voigt_before
    lines_before = mod._cache['lines']
    # in-place edit of the line strengths: seen, and the result follows
    case['lines']['gf'] *= 2.0
    doubled = _dropin_call(hip, case, profile, psize, pindex, 1)
    assert mod._cache['lines'] is not lines_before
    nz = base != 0
    np.testing.assert_allclose(doubled[nz], 2.0 * base[nz], rtol=1e-12)
    # in-place edit of the table
    profile *= 3.0
    tripled = _dropin_call(hip, case, profile, psize, pindex, 1)
    assert mod._cache['voigt'] is not voigt_before
    np.testing.assert_allclose(tripled[nz], 6.0 * base[nz], rtol=1e-12)
    mod.invalidate()
    assert not mod._seen and not mod._cache


def test_extinction_cache_sees_a_single_edited_line(hip):
    """with the default thresholds the line-list arrays are hashed in full on
    every call, so an in-place edit of ONE element that no strided probe would visit (here
    element 12 345 of gf, then two swapped lines) changes the result like in the reference,
    which re-reads its inputs on every call.  A read-only huge buffer is probed like a writeable one."""
    mod = hip._extcoeff
    mod.invalidate()
    case, profile, psize, pindex = _dropin_case()
    ln = case['lines']
    probed = set(np.linspace(0, ln['gf'].size - 1, mod._PROBE_ELEMS).astype(int)[1:-1])
    i = next(j for j in range(12345, 13000) if j not in probed)
    base = _dropin_call(hip, case, profile, psize, pindex, 1)
    ln['gf'][i] *= 1e6                                    # one line, in place
    edited = _dropin_call(hip, case, profile, psize, pindex, 1)
    assert not np.array_equal(edited, base)
    ln['gf'][i] /= 1e6
    assert np.array_equal(_dropin_call(hip, case, profile, psize, pindex, 1), base)
    j = i + 1                                             # swap the strengths of two lines
    ln['gf'][[i, j]] = ln['gf'][[j, i]]
    ln['elow'][[i, j]] = ln['elow'][[j, i]]
    swapped = _dropin_call(hip, case, profile, psize, pindex, 1)
    assert not np.array_equal(swapped, base)
    # read-only buffers are probed like any other: an unfreeze-edit-refreeze at
    # the same address, or another read-only array at a recycled address, must change the key
    big = np.arange(5 << 20, dtype=float)                 # 40 MiB > 32 MiB
    big.setflags(write=False)
    view = big[:]
    assert mod._frozen(view) and not mod._frozen(np.arange(4.0)[:2])
    k1 = mod._content_key(view)
    calls, full = [], []
    real, real_digest = mod._probe, mod._digest
    mod._probe = lambda a: (calls.append(1), real(a))[1]
    mod._digest = lambda a: (full.append(1), real_digest(a))[1]
    try:
        assert mod._content_key(view) == k1 and len(calls) == 1 and not full   # probe, no re-hash
        big.setflags(write=True)
        big *= 2.0                                        # (every element: the probe sees it)
        big.setflags(write=False)
        k2 = mod._content_key(big[:])
        assert k2 != k1 and full
    finally:
        mod._probe, mod._digest = real, real_digest
    mod.invalidate()


def test_extinction_cache_without_xxhash(hip, monkeypatch):
    """xxhash is optional: without it the content hashes come from hashlib (same results)."""
    mod = hip._extcoeff
    mod.invalidate()
    case, profile, psize, pindex = _dropin_case()
    want = _dropin_call(hip, case, profile, psize, pindex, 2)
    mod.invalidate()
    monkeypatch.setattr(mod, '_xxhash', None)
    assert np.array_equal(_dropin_call(hip, case, profile, psize, pindex, 2), want)
    assert np.array_equal(_dropin_call(hip, case, profile, psize, pindex, 2), want)
    mod.invalidate()
