"""Error behaviour of the C ABI on a GPU box: every entry validates shapes and pointers
before launching (a faulting kernel can take the whole node down), returns a status and
leaves a message in pb_last_error(); nothing falls back to a CPU path."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


def test_column_entries_reject_bad_arguments(eng):
    from pyratbay_amd._capi import call, PbError
    d = eng.dev(np.zeros((4, 8)))
    ideep = eng.dev(np.zeros(8, np.int32), dtype=__import__('torch').int32)
    radius = eng.dev(np.linspace(4.0, 1.0, 4))
    spec = eng.dev(np.zeros(8))
    s = eng._stream()
    p = eng._ptr
    with pytest.raises(PbError, match='itop out of range'):
        call('pb_transmission', p(spec), p(d), p(ideep), p(radius), 7, 1.0, 4, 8, s)
    with pytest.raises(PbError, match='null pointer'):
        call('pb_transmission', None, p(d), p(ideep), p(radius), 0, 1.0, 4, 8, s)
    with pytest.raises(PbError, match='deck_itop'):
        call('pb_transmission_deck', p(spec), p(d), p(ideep), p(radius), 0, 1.0, 9, 2.0, 4, 8, s)
    with pytest.raises(PbError, match='ibottom'):
        call('pb_optical_depth_transit', p(d), p(ideep), p(d), p(radius), 0, 9, 10.0, 4, 8, s)
    with pytest.raises(PbError, match='rtop out of range'):
        call('pb_two_stream', p(d), p(d), p(d), p(spec), p(radius), None, None, 5, 4, 8, s)
    with pytest.raises(PbError, match='nmu'):
        call('pb_emission_flux', p(spec), None, p(d), p(ideep), p(spec), p(radius), p(radius),
             p(radius), 0, 0, 4, 8, s)
    # empty spectra are fine and touch nothing
    call('pb_transmission', None, None, None, None, 0, 1.0, 4, 0, s)


def test_continuum_limits(eng):
    from pyratbay_amd._capi import call, PbError, hptr
    ec = eng.dev(np.zeros((2, 4)))
    temp = eng.dev(np.array([500.0, 600.0]))
    n = np.zeros(5, np.int32)
    ptrs = (C.c_void_p * 5)()
    with pytest.raises(PbError, match='at most 4 CIA'):
        call('pb_continuum', eng._ptr(ec), None, eng._ptr(temp), 2, 4, 0, None, None, 5,
             C.cast(ptrs, C.c_void_p), C.cast(ptrs, C.c_void_p), hptr(n), hptr(n), hptr(n),
             eng._ptr(temp), None, None, None, eng._stream())
    with pytest.raises(PbError, match='rank-1 terms without arrays'):
        call('pb_continuum', eng._ptr(ec), None, eng._ptr(temp), 2, 4, 2, None, None, 0, None,
             None, None, None, None, None, None, None, None, eng._stream())
    with pytest.raises(PbError, match='at most 8 lines'):
        wn0 = np.zeros(9)
        call('pb_alkali_cross_section', eng._ptr(ec), eng._ptr(temp), eng._ptr(temp),
             eng._ptr(temp), eng._ptr(temp), 1.0, 1.0, 1.0, 1.0, 1.0, hptr(wn0), hptr(wn0), 9,
             None, 2, 4, eng._stream())


def test_lbl_plan_rejects_bad_calls(eng):
    from pyratbay_amd import synth
    from pyratbay_amd._capi import PbError
    case = synth.lbl_case(257, 3, 100, wnosamp=12, nlor=8, ndop=4, extent=30.0, cutoff=2.0)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 12)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], 1, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                  vg['cutoff'], 1e-30, max_layers=2)
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])
    with pytest.raises(PbError):                      # 3 layers into a plan sized for 2
        lbl.extinction(t, d, z)
    with pytest.raises(PbError):                      # shard beyond the grid
        lbl.extinction(t[:2], d[:2], z[:, :2].contiguous(), wbegin=200, wcount=100)
    with pytest.raises(KeyError):
        lbl.set_gather_mode('fastest')
    ok = lbl.extinction(t[:2].contiguous(), d[:2].contiguous(), z[:, :2].contiguous())
    assert ok.shape == (2, 1, 257) and bool((ok >= 0).all())


def test_ordered_batches_drop_indices_outside_the_grid(eng):
    """pb_transit_spectrum_ordered / pb_emission_flux_ordered scatter every column to the grid
    index the caller gives; an index outside the grid (a column_d that is not a permutation) is
    dropped by the kernels, not written out of bounds: the other columns are those of the
    unordered call, bit for bit, and the spectrum array is touched nowhere else."""
    import torch
    import cases
    rng = np.random.default_rng(3)
    L, W, nw = 24, 333, 2
    c = cases.column_case(seed=4, nlayers=L, nwave=W)
    ecs = np.array([c['ec'] * 10.0**rng.uniform(-1, 1) for _ in range(nw)])
    radius = np.array([c['radius'] for _ in range(nw)])
    ec_d, rad_d = eng.dev(ecs), eng.dev(radius)
    path = eng.transit_path_device(rad_d, 0)
    want = eng.transit_spectrum_batch(ec_d, path, rad_d, c['rstar'], 0, L, 10.0)
    order = torch.arange(W, device='cuda', dtype=torch.int32)
    bad = order.clone()
    bad[7], bad[100], bad[W - 1] = -1, W, 2**30
    keep = torch.ones(W, dtype=torch.bool, device='cuda')
    keep[[7, 100, W - 1]] = False
    # (a guard band around the spectrum: the library writes through raw pointers)
    buf = torch.full((nw * W + 2 * 4096,), 7.0, dtype=torch.float64, device='cuda')
    spec = buf[4096:4096 + nw * W].view(nw, W)
    nwork = eng._capi.lib().pb_transit_work_doubles(L, 0, L, W, nw)
    work = torch.empty(nwork, dtype=torch.float64, device='cuda')
    eng.call('pb_transit_spectrum_ordered', eng._ptr(spec), eng._ptr(ec_d), eng._ptr(path),
             eng._ptr(rad_d), eng._ptr(bad), float(c['rstar']), 0, L, 10.0, L, W, nw,
             eng._ptr(work), eng._stream())
    assert torch.equal(spec[:, keep], want[:, keep])
    assert bool((spec[:, ~keep] == 7.0).all())
    assert bool((buf[:4096] == 7.0).all()) and bool((buf[4096 + nw * W:] == 7.0).all())
    # emission
    temps = eng.dev(np.array([c['temp'] for _ in range(nw)]))
    intervals = (rad_d[:, :-1] - rad_d[:, 1:]).contiguous()
    mu, wts = eng.default_quadrature()
    mu_d, w_d, wn_d = eng.dev(mu), eng.dev(wts), eng.dev(c['wn'])
    want_e = eng.emission_flux_batch(ec_d, intervals, wn_d, temps, mu_d, w_d, 0, L, 10.0)
    buf.fill_(7.0)
    eng.call('pb_emission_flux_ordered', eng._ptr(spec), eng._ptr(ec_d), eng._ptr(intervals),
             eng._ptr(wn_d), eng._ptr(temps), eng._ptr(mu_d), eng._ptr(w_d), eng._ptr(bad),
             len(mu), 10.0, 0, L, L, W, nw, eng._stream())
    assert torch.equal(spec[:, keep], want_e[:, keep])
    assert bool((spec[:, ~keep] == 7.0).all())
    assert bool((buf[:4096] == 7.0).all()) and bool((buf[4096 + nw * W:] == 7.0).all())
