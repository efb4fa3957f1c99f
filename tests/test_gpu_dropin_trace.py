"""Fixture G17 (tests/golden/make_golden_trace.py): every call the reference's own Python layer
made into its native modules during pb.run() (transit, emission, emission with quadrature = 3) on
the mock H2O list -- recorded with each argument's dtype, byte strides and flags -- replayed
through the drop-in modules `pyratbay_amd.lib.*` with arrays rebuilt at the recorded dtypes and
strides (int64 index arrays read as C int, `ind.h:7-37`; the non-contiguous iso_pf[:, layer]
column of pyrat/extinction.py:189; int32 `ideep` in transit and int64 in emission geometry,
opacity/optic_depth.py:89-136; blackbody_wn_2D without its optional `last`).

Compared per call: the return value and the post-call state of every argument the reference
changed in place -- floats at 1e-10 (the Voigt table: 2e-12 off the x87 series), integers
exactly, zero patterns of the extinction rows identical."""
import json

import numpy as np
import pytest

RTOL = 1e-10


@pytest.fixture(scope='module')
def trace(golden):
    g = golden('g17_call_trace')
    meta = json.loads(str(g['trace_json']))
    return g, meta


def rebuild(desc, g):
    """The argument as the reference's caller passed it: same dtype, shape and byte strides (a
    strided view into a larger buffer when it was not contiguous), same writeable flag."""
    kind = desc['kind']
    if kind == 'none':
        return None
    if kind in ('int', 'float', 'bool'):
        return desc['value'] if kind != 'float' else float.fromhex(desc['hex'])
    if kind == 'list':
        return [rebuild(d, g) for d in desc['items']]
    values = g['arr_' + desc['values']]
    dtype = np.dtype(desc['dtype'])
    shape, strides = tuple(desc['shape']), tuple(desc['strides'])
    assert values.dtype == dtype and values.shape == shape
    if desc['c_contiguous'] or values.size == 0:
        a = values.copy()
    else:
        assert all(s >= 0 for s in strides)
        extent = sum((n - 1) * s for n, s in zip(shape, strides)) + dtype.itemsize
        base = np.full(extent, 0xA5, np.uint8)              # (junk between the elements)
        a = np.ndarray(shape, dtype, buffer=base, strides=strides)
        a[...] = values
        assert a.strides == strides and not a.flags.c_contiguous
    if not desc['writeable']:
        a.setflags(write=False)
    return a


def compare(got, want_desc, g, what):
    kind = want_desc['kind']
    if kind == 'none':
        assert got is None, what
    elif kind == 'int':
        assert int(got) == want_desc['value'], what
    elif kind == 'float':
        np.testing.assert_allclose(float(got), want_desc['value'], rtol=RTOL, err_msg=what)
    else:
        want = g['arr_' + want_desc['values']]
        got = np.asarray(got)
        assert got.shape == want.shape, what
        if want.dtype.kind in 'iu':
            assert np.array_equal(got, want), what
        else:
            assert got.dtype == want.dtype, what
            np.testing.assert_allclose(got, want, rtol=RTOL, atol=0, err_msg=what)


def test_trace_covers_the_path(trace):
    _, meta = trace
    counts = meta['counts']
    for fn in ('vprofile.grid', '_extcoeff.extinction', '_trapezoid.optdepth',
               '_trapezoid.trapezoid2D', '_trapezoid.plane_parallel_optical_depth',
               '_blackbody.blackbody_wn_2D', '_trapezoid.intensity', 'cutils.ediff',
               '_indices.ifirst', '_blackbody.blackbody_wn'):
        assert counts.get(fn, 0) > 0, fn
    ext = [c for c in meta['calls'] if c['function'] == 'extinction']
    # what no hand-written loop of the test-suite passed before: int64 size / index / isotope
    # arrays, a strided partition-function column, verb = -10
    assert ext[0]['args'][2]['dtype'] == '<i8' and ext[0]['args'][20]['dtype'] == '<i8'
    assert not ext[0]['args'][15]['c_contiguous'] and ext[0]['args'][24]['value'] < 0
    pp = [c for c in meta['calls'] if c['function'] == 'plane_parallel_optical_depth'][0]
    od = [c for c in meta['calls'] if c['function'] == 'optdepth'][0]
    assert pp['args'][1]['dtype'] == '<i8' and od['args'][3]['dtype'] == '<i4'


@pytest.mark.gpu
def test_replay_reference_callers_through_the_dropins(trace):
    from pyratbay_amd import engine
    engine.require_gpu()
    import pyratbay_amd.lib as hip
    g, meta = trace
    hip._extcoeff.invalidate()
    done = {}
    for n, c in enumerate(meta['calls']):
        mod = getattr(hip, c['module'], None)
        assert mod is not None, f"no drop-in for pyratbay.lib.{c['module']}"
        fn = getattr(mod, c['function'])
        args = [rebuild(a, g) for a in c['args']]
        what = f"call {n}: {c['run']} {c['module']}.{c['function']} (#{c['nth']})"
        ret = fn(*args)
        compare(ret, c['return'], g, what + ' return value')
        for i, a in enumerate(c['args']):
            if a['kind'] != 'array':
                continue
            key = c['after'].get(str(i), a['values'])        # changed in place, or untouched
            want = g['arr_' + key]
            got = np.asarray(args[i])
            if want.dtype.kind in 'iu':
                assert np.array_equal(got, want), f'{what}: argument {i} after the call'
            else:
                if c['function'] == 'extinction' and i == 0:
                    assert np.array_equal(got == 0, want == 0), f'{what}: zero pattern'
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=0,
                                           err_msg=f'{what}: argument {i} after the call')
        k = f"{c['module']}.{c['function']}"
        done[k] = done.get(k, 0) + 1
    hip._extcoeff.invalidate()
    assert done == {k: sum(1 for c in meta['calls'] if f"{c['module']}.{c['function']}" == k)
                    for k in done}
    print('replayed', sum(done.values()), 'recorded calls:', done)
