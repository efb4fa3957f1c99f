"""pyratbay.lib._extcoeff (src_c/_extcoeff.c) on the GPU.

`extinction` keeps the reference's one-layer-per-call signature; the Voigt table and the
line list are uploaded once and cached on the device between calls (keyed on the
caller's arrays), so the per-layer loop of pyratbay/pyrat/extinction.py:170-213 does not
re-upload them."""
import numpy as np
import torch

from .. import engine
from . import _np

_cache = {}
_seen = {}                         # buffer identity -> (content digest, probe digest)

_FULL_HASH_BYTES = 256 << 20       # arrays up to this size are hashed in full
# arrays up to this size are hashed in full on EVERY call: the line-list arrays (lwn, elow, gf,
# lID: 8 MB each at 1e6 lines, ~0.5 ms at xxh3's 15 GB/s), the width grids, the species data.  The
# reference re-reads its inputs on every call, so an in-place edit of ANY element of them (a few
# gf values scaled, two lines swapped) must change the result -- a strided probe cannot promise
# that.  Only the genuinely huge arrays (`own`: 144 MB at C2, the Voigt table: GBs) are
# identified by buffer identity + a probe.
_SMALL_BYTES = 32 << 20
_PROBE_ELEMS = 8192                # elements of the per-call probe of a known buffer (64 KiB)
_SEEN_MAX = 64

try:                               # optional dependency (an offline wheel in this image)
    import xxhash as _xxhash
except ImportError:                # pragma: no cover - exercised by monkeypatching in the tests
    _xxhash = None


def _hash(*buffers):
    """64-bit content hash of byte buffers: xxh3 (~15 GB/s) when the xxhash package is
    importable, else hashlib.blake2b (~1 GB/s; same semantics, slower first call)."""
    if _xxhash is not None:
        h = _xxhash.xxh3_64()
        for b in buffers:
            h.update(b)
        return h.intdigest()
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for b in buffers:
        h.update(b)
    return int.from_bytes(h.digest(), 'little')


def _bytes(a):
    return np.ascontiguousarray(a).reshape(-1).view(np.uint8)


def _digest(a):
    """Content hash of an array: every byte up to 256 MiB; beyond that -- only the Voigt table
    gets there -- 2^20 evenly spaced elements plus both ends."""
    a = np.asarray(a)
    if a.nbytes <= _FULL_HASH_BYTES:
        return _hash(_bytes(a))
    flat = a.reshape(-1)
    step = max(1, flat.size >> 20)
    return _hash(_bytes(flat[::step]), _bytes(flat[:8192]), _bytes(flat[-8192:]))


def _probe(a):
    """Hash of _PROBE_ELEMS evenly spaced elements + the first and last 64 (no copy of `a`)."""
    n = a.size
    idx = np.linspace(0, n - 1, _PROBE_ELEMS).astype(np.intp)
    flat = a.flat
    return _hash(_bytes(flat[idx]), _bytes(flat[np.arange(64)]), _bytes(flat[np.arange(n - 64, n)]))


def _frozen(a):
    """True when nothing can write to the buffer through NumPy: the array and every array it
    is a view of are read-only (a read-only view of a writeable array is not frozen)."""
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None or isinstance(a, bytes)


def _content_key(a):
    """(shape, content digest) of a caller's array, cheap for a buffer seen before.

    The reference re-reads its inputs on every call; a device cache must notice when they
    change.  Hashing everything in full on every layer call cost 8.5 ms per call at C2 (`own`
    alone is 144 MB).  Now: arrays of up to 32 MiB -- every line-list array up to 4e6 lines --
    are hashed in full every time (an edit of a few gf values must be seen).
    A larger buffer is identified by (address, shape, strides, dtype, writeable); the first
    time it is hashed in full, afterwards only a 64-KiB strided probe of it is -- when the
    probe differs (an in-place edit, or another array at a recycled address) the full hash is
    taken again.  The probe is taken for read-only buffers too (a freed
    read-only table whose address the allocator hands to another read-only array of the same
    shape -- large mmap'd blocks are recycled that way -- or setflags(write=True), an edit,
    setflags(write=False) must be seen; identity alone cannot promise either).  An in-place edit
    of such a huge array (`own`, the Voigt table) that misses every probed element is not seen:
    call invalidate() after it."""
    a = np.asarray(a)
    if a.nbytes <= _SMALL_BYTES:
        return a.shape, _hash(_bytes(a))
    ident = (a.__array_interface__['data'][0], a.shape, a.strides, a.dtype.str, _frozen(a))
    ent = _seen.get(ident)
    probe = _probe(a)
    if ent is not None and ent[1] == probe:
        return a.shape, ent[0]
    digest = _digest(a)
    if len(_seen) >= _SEEN_MAX:
        _seen.pop(next(iter(_seen)))
    _seen[ident] = (digest, probe)
    return a.shape, digest


def _key(*arrays):
    return tuple(_content_key(a) for a in arrays)


def invalidate():
    """Drop the device copies of the Voigt table, the line list and the plan: the next
    extinction() call uploads the caller's arrays again."""
    for name in ('lbl', 'lines', 'voigt'):
        old = _cache.pop(name, None)
        if old is not None:
            old.close()
    _cache.clear()
    _seen.clear()


def _voigt(profile, psize, pindex, lorentz, doppler, osamp):
    key = ('voigt', osamp) + _key(profile, psize, pindex, lorentz, doppler)
    if _cache.get('voigt_key') != key:
        old = _cache.pop('voigt', None)
        if old is not None:
            old.close()
        _cache['voigt'] = engine.VoigtTable.from_flat(
            _np.f64(profile), _np.read_int(psize), _np.read_int(pindex), _np.f64(lorentz),
            _np.f64(doppler), osamp, keep_flat=False)
        _cache['voigt_key'] = key
        _cache.pop('lbl_key', None)
    return _cache['voigt']


def _lines(lwn, elow, gf, lID, niso, own):
    key = ('lines', niso) + _key(lwn, elow, gf, lID, own)
    if _cache.get('lines_key') != key:
        old = _cache.pop('lines', None)
        if old is not None:
            old.close()
        _cache['lines'] = engine.LineList(_np.f64(lwn), _np.f64(elow), _np.f64(gf),
                                          _np.read_int(lID), niso, _np.f64(own))
        _cache['lines_key'] = key
        _cache.pop('lbl_key', None)
    return _cache['lines']


def extinction(ext, profile, psize, pindex, lorentz, doppler, wn, own, divisors,
               moldensity, molrad, molmass, isoimol, isomass, isoratio, isoz, isoiext,
               lwn, elow, gf, lID, cutoff, ethresh, temp, verb, add=0, resolution=0):
    """extinction(ext, profile, psize, pindex, lorentz, doppler, wn, own, divisors,
    moldensity, molrad, molmass, isoimol, isomass, isoratio, isoz, isoiext, lwn, elow,
    gf, lID, cutoff, ethresh, temp, verb[, add, resolution]) -> 1

    Writes ext[nextinct, nwave] in place: assigned in constant-step mode, accumulated in
    resolution mode (linterp), exactly like _extcoeff.c:320-332."""
    wn = _np.f64(wn)
    own = _np.f64(own)
    div = _np.read_int(divisors)
    osamp = int(div[-1])
    niso = len(np.atleast_1d(isomass))
    voigt = _voigt(profile, psize, pindex, lorentz, doppler, osamp)
    lines = _lines(lwn, elow, gf, lID, niso, own)
    iext = _np.read_int(isoiext)
    key = ('lbl',) + _key(wn, div, molrad, molmass, isoimol, isomass, isoratio) + (
        float(cutoff), int(bool(resolution)), int(iext.max()))
    if _cache.get('lbl_key') != key:
        old = _cache.pop('lbl', None)
        if old is not None:
            old.close()
        _cache['lbl'] = engine.LBL(voigt, lines, wn, div, _np.f64(molrad), _np.f64(molmass),
                                   _np.read_int(isoimol), _np.f64(isomass),
                                   _np.f64(isoratio), iext, float(cutoff), float(ethresh),
                                   resolution=bool(resolution), max_layers=1)
        _cache['lbl_key'] = key
        _cache.pop('iext', None)
    lbl = _cache['lbl']
    if _cache.get('iext') != iext.tobytes():
        lbl.set_isoiext(iext)
        _cache['iext'] = iext.tobytes()
    lbl.set_ethresh(float(ethresh))
    rows = 1 if add else lbl.nrows_sep
    if ext.shape[0] < rows or ext.shape[1] != len(wn):
        raise ValueError(f'ext has shape {ext.shape}, expected ({rows}, {len(wn)})')
    if resolution:
        # linterp ACCUMULATES into ext (_extcoeff.c:320-326): the caller's values go up
        out = _np.dev(np.ascontiguousarray(ext[:rows], dtype=np.float64)).reshape(1, rows, len(wn))
    else:
        out = torch.empty((1, rows, len(wn)), dtype=torch.float64, device='cuda')
    temp_d = _np.dev(np.array([temp], float))
    dens_d = _np.dev(_np.f64(moldensity).reshape(1, -1))
    z_d = _np.dev(_np.f64(isoz).reshape(-1, 1))
    lbl.extinction(temp_d, dens_d, z_d, add=bool(add), out=out)
    ext[:rows] = _np.host(out)[0]
    return 1


def _interp(extinction_, etable, ttable, temperatures, density, lay1, lay2, per_mol):
    ext_d = _np.dev(_np.f64(extinction_))
    et = _np.dev(_np.f64(etable))
    engine.interp_ec(ext_d, et, _np.dev(_np.f64(ttable)), _np.dev(_np.f64(temperatures)),
                     _np.dev(_np.f64(density)), int(lay1), int(lay2), per_mol)
    extinction_[...] = _np.host(ext_d)
    return 1


def interp_ec(extinction, etable, ttable, temperatures, density, lay1, lay2):
    """interp_ec(extinction, etable, ttable, temperatures, density, lay1, lay2) -> 1
    (src_c/_extcoeff.c:367-418); accumulates into extinction[nlayers, nwave]."""
    return _interp(extinction, etable, ttable, temperatures, density, lay1, lay2, False)


def interp_ec_per_mol(extinction, etable, ttable, temperatures, density, lay1, lay2):
    """Same with extinction[nmol, nlayers, nwave] (src_c/_extcoeff.c:422-472)."""
    return _interp(extinction, etable, ttable, temperatures, density, lay1, lay2, True)
