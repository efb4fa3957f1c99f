"""Device-resident front-end of the hot path (host side above the C ABI).

Mirrors the call structure of the reference's Python layer for this path

    Voigt()                         pyratbay/pyrat/voigt.py:46-149
    extinction() per-layer loop     pyratbay/pyrat/extinction.py:129-213
    optical_depth()                 pyratbay/opacity/optic_depth.py:16-144
    transmission()/plane_parallel_rt()  pyratbay/spectrum/radiative_transfer.py:23-138
    spectrum()                      pyratbay/pyrat/spectrum.py:333-385

but keeps every array ([layer][wavenumber], row-major) in HBM between the stages:
ec, depth and B never travel to the host unless asked for.  torch tensors are used
only as owners of device memory; all arithmetic happens in libpbhip.so.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi
from ._capi import call, hptr, f64h, i32h


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """The current HIP stream of torch as the `void *stream` of the C ABI.  Every library call
    asks for it; torch.cuda.current_stream() builds a Stream object through several Python layers
    (~8 us, a fifth of the host's submission time of a rank-size spectrum), the raw getters are
    one C call each."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE_STREAMS = []


def side_streams(n):
    """The process-wide side streams 0 .. n-1 (created once, shared by every pipeline object).
    HIP maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4) in the order
    they are created; two streams on one queue run their kernels strictly one after the other.
    A process that makes fresh streams for every SpectrumPipeline / ShardPipeline soon has two
    "concurrent" contexts on the same queue (seen in a kernel trace: the second of two pipelines
    of one process ran fully serialised).  Re-using the same few streams keeps the mapping the
    one the first pipeline got."""
    while len(_SIDE_STREAMS) < n:
        _SIDE_STREAMS.append(torch.cuda.Stream())
    return _SIDE_STREAMS[:n]


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def dev(a, dtype=torch.float64, device=None):
    """Host array -> contiguous device tensor of the ABI's element type."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device or 'cuda', dtype=dtype).contiguous()
    np_dtype = {torch.float64: np.float64, torch.int32: np.int32,
                torch.int64: np.int64}[dtype]
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np_dtype)).to(device or 'cuda')


def require_gpu():
    if not torch.cuda.is_available():
        raise _capi.PbError('no GPU visible: the HIP path has no CPU fallback')
    _capi.lib()


# --------------------------------------------------------------------------
# Stage timers and profiler ranges
# --------------------------------------------------------------------------
class StageTimer:
    """The reference's `pyrat.timestamps` for the device path (pyrat_obj.py:203-214 with the
    Timer of tools/tools.py:832-843): seconds spent in each named stage since the previous
    mark, measured with HIP events on the launch stream and resolved lazily -- start() and
    mark() only enqueue an event, read() waits for the last one.  Every stage is also a rocTX
    range (rocprofv3 --marker-trace)."""

    def __init__(self, max_stages=8):
        self._h = C.c_void_p()
        call('pb_timer_create', C.byref(self._h), int(max_stages))

    def start(self, first_stage=None):
        call('pb_timer_start', self._h, None if first_stage is None else first_stage.encode(),
             _stream())

    def mark(self, name, next_stage=None):
        call('pb_timer_mark', self._h, name.encode(),
             None if next_stage is None else next_stage.encode(), _stream())

    def read(self):
        """{stage: seconds} of the stages marked since the last start(), in order."""
        n = C.c_int(0)
        call('pb_timer_count', self._h, C.byref(n))
        out = {}
        buf = C.create_string_buffer(64)
        for i in range(n.value):
            sec = C.c_double(0)
            call('pb_timer_read', self._h, i, buf, 64, C.byref(sec))
            key = buf.value.decode()
            out[key] = out.get(key, 0.0) + sec.value
        return out

    def close(self):
        if self._h:
            call('pb_timer_destroy', self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class profiler_range:
    """with profiler_range('all_gather'): ...  -- a rocTX range (no-op without the marker
    library; rocprofv3 --marker-trace shows it beside the kernels)."""

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        call('pb_range_push', self.name)
        return self

    def __exit__(self, *exc):
        call('pb_range_pop')
        return False


# --------------------------------------------------------------------------
# Voigt table
# --------------------------------------------------------------------------
class VoigtTable:
    """Grid of Voigt profiles resident on the device (vprofile.grid, voigt.py:133-149)."""

    def __init__(self, handle, nlor, ndop, lorentz, doppler, osamp):
        self._h = handle
        self.nlor, self.ndop, self.osamp = nlor, ndop, osamp
        self.lorentz, self.doppler = lorentz, doppler
        size = np.zeros((nlor, ndop), np.int32)
        index = np.zeros((nlor, ndop), np.int32)
        n = C.c_int64(0)
        call('pb_voigt_meta', self._h, hptr(size), hptr(index), C.byref(n))
        self.size, self.index, self.nprofile = size, index, n.value

    @classmethod
    def build(cls, lorentz, doppler, size, ownstep, osamp, keep_flat=False):
        require_gpu()
        lorentz, doppler = f64h(lorentz), f64h(doppler)
        size = i32h(size)
        h = C.c_void_p()
        call('pb_voigt_create', C.byref(h), hptr(lorentz), len(lorentz), hptr(doppler),
             len(doppler), hptr(size), float(ownstep), int(osamp), int(keep_flat), _stream())
        return cls(h, len(lorentz), len(doppler), lorentz, doppler, int(osamp))

    @classmethod
    def from_flat(cls, profile, size, index, lorentz, doppler, osamp, keep_flat=False):
        require_gpu()
        profile, lorentz, doppler = f64h(profile), f64h(lorentz), f64h(doppler)
        size, index = i32h(size), i32h(index)
        h = C.c_void_p()
        call('pb_voigt_from_flat', C.byref(h), hptr(profile), profile.size, hptr(lorentz),
             len(lorentz), hptr(doppler), len(doppler), hptr(size), hptr(index), int(osamp),
             int(keep_flat), _stream())
        return cls(h, len(lorentz), len(doppler), lorentz, doppler, int(osamp))

    def flat(self, out=None):
        """The table in the reference's layout (what vprofile.grid fills), on the host."""
        if out is None:
            out = np.zeros(self.nprofile)
        assert out.dtype == np.float64 and out.flags.c_contiguous
        call('pb_voigt_flat_to_host', self._h, hptr(out), out.size)
        return out

    @property
    def device_bytes(self):
        return call('pb_voigt_device_bytes', self._h)

    def close(self):
        if self._h:
            call('pb_voigt_destroy', self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# Line list
# --------------------------------------------------------------------------
class LineList:
    """Line transitions + the co-add groups of _extcoeff.c:243-262, on the device."""

    def __init__(self, lwn, elow, gf, lid, niso, own):
        require_gpu()
        lwn, elow, gf, lid, own = f64h(lwn), f64h(elow), f64h(gf), i32h(lid), f64h(own)
        self.nlines, self.niso = len(lwn), int(niso)
        self._h = C.c_void_p()
        call('pb_lines_create', C.byref(self._h), hptr(lwn), hptr(elow), hptr(gf), hptr(lid),
             len(lwn), int(niso), hptr(own), len(own), float(own[0]), float(own[1] - own[0]))
        st = (C.c_int64 * 3)()
        call('pb_lines_stats', self._h, C.byref(st))
        self.ninrange, self.ngroups, self.nadd = st[0], st[1], st[2]
        flag = C.c_int(0)
        call('pb_lines_grouped_on_device', self._h, C.byref(flag))
        self.grouped_on_device = bool(flag.value)

    def groups(self):
        """(first, count, iown)[ngroups] and iso_gstart[niso + 1] of the co-add groups."""
        first = np.zeros(self.ngroups, np.int32)
        count = np.zeros(self.ngroups, np.int32)
        iown = np.zeros(self.ngroups, np.int32)
        start = np.zeros(self.niso + 1, np.int64)
        call('pb_lines_groups', self._h, hptr(first), hptr(count), hptr(iown), hptr(start))
        return first, count, iown, start

    def close(self):
        if self._h:
            call('pb_lines_destroy', self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# Partition functions Z_i(T)
# --------------------------------------------------------------------------
class PartitionTable:
    """The partition-function tables of a TLI file's databases on the device, evaluated at a
    temperature profile the way Line_By_Line does on every extinction call
    (pyratbay/pyrat/line_by_line.py:156-158: interp1d(db.temp, db.iso_pf[j], kind='slinear');
    :219-222: iso_pf[i] = iso_pf_interp[i](temperature)).  `databases`: the header dicts of
    pyratbay_amd.tli.read_tli (keys temperatures[ntemp], partition[niso, ntemp]); isotopes are
    numbered over the concatenated databases, as the line list's isotope index is."""

    def __init__(self, databases):
        require_gpu()
        self.tables = []
        self.niso = 0
        for db in databases:
            t = np.ascontiguousarray(db['temperatures'], float)
            pf = np.ascontiguousarray(np.atleast_2d(db['partition']), float)
            if t.size < 2 or pf.shape[1] != t.size or np.any(np.diff(t) <= 0):
                raise ValueError('partition-function table: temperatures must be strictly '
                                 'ascending (at least two) and match partition[niso, ntemp]')
            self.tables.append((dev(t), dev(pf), self.niso, pf.shape[0]))
            self.niso += pf.shape[0]
        self._nbad = torch.zeros(1, dtype=torch.int32, device='cuda')

    def evaluate(self, temp, out=None, check=True):
        """temp: device tensor of any shape [...] -> Z[niso, ...] (float64, device).  check=True
        waits for the kernel and raises ValueError for a temperature outside a table, like
        interp1d; check=False leaves NaN in those places and does not synchronise (a batch of
        walkers whose out-of-range members the caller rejects by their NaN spectra)."""
        temp = temp.contiguous()
        n = temp.numel()
        if out is None:
            out = torch.empty((self.niso,) + tuple(temp.shape), dtype=torch.float64,
                              device=temp.device)
        assert out.is_contiguous() and out.numel() == self.niso * n
        if check:
            self._nbad.zero_()
        for ttab, pf, first, niso in self.tables:
            call('pb_iso_partition', out.data_ptr() + 8 * first * n, n, 1, _ptr(temp), n,
                 _ptr(ttab), ttab.numel(), _ptr(pf), niso, _ptr(self._nbad) if check else None,
                 _stream())
        if check and int(self._nbad.item()) > 0:
            lo = max(float(t[0][0]) for t in self.tables)
            hi = min(float(t[0][-1]) for t in self.tables)
            raise ValueError('A value in the temperature profile lies outside the partition-'
                             f'function tables ({lo} - {hi} K)')
        return out


# --------------------------------------------------------------------------
# LBL extinction
# --------------------------------------------------------------------------
class LBL:
    """All-layer line-by-line extinction (the loop of extinction.py:170-213)."""

    def __init__(self, voigt, lines, wn, divisors, molrad, molmass, isoimol, isomass,
                 isoratio, isoiext, cutoff, ethresh, resolution=False, max_layers=256):
        self.voigt, self.lines = voigt, lines       # keep the handles alive
        wn = f64h(wn)
        self.nwave = len(wn)
        self.nmol, self.niso = len(molmass), len(isomass)
        isoiext = i32h(isoiext)
        self.nrows_sep = max(1, int(isoiext.max()) + 1)
        self.max_layers = max_layers
        self._h = C.c_void_p()
        args = [f64h(molrad), f64h(molmass), i32h(isoimol), f64h(isomass), f64h(isoratio)]
        div = i32h(divisors)
        call('pb_lbl_create', C.byref(self._h), voigt._h, lines._h, hptr(wn), len(wn),
             hptr(div), len(div), hptr(args[0]), hptr(args[1]), self.nmol,
             hptr(args[2]), hptr(args[3]), hptr(args[4]), hptr(isoiext), self.niso,
             float(cutoff), float(ethresh), int(bool(resolution)), int(max_layers))
        self.resolution = bool(resolution)
        self.gather_mode = 'auto'

    def set_isoiext(self, isoiext):
        isoiext = i32h(isoiext)
        call('pb_lbl_set_isoiext', self._h, hptr(isoiext))

    def set_ethresh(self, ethresh):
        call('pb_lbl_set_ethresh', self._h, float(ethresh))

    GATHER = {'auto': 0, 'global': 1, 'staged': 2, 'resident': 3, 'scatter': 4, 'rounds': 5,
              'dynamic': 6, 'wave': 7}

    def set_gather_mode(self, mode):
        """'auto' | 'global' | 'staged' | 'resident'; 'dynamic' (`resolution` plans: the layers'
        dynamic grids through constant-step sub-plans).  'scatter', 'rounds' and 'wave' are
        measured dead ends that only the experiments build of the library carries
        (libpbhip_exp.so, _capi.experiments()); the default library refuses them.  See pbhip.h:
        pb_lbl_set_gather_mode."""
        call('pb_lbl_set_gather_mode', self._h, self.GATHER[mode])
        self.gather_mode = mode

    def set_record_budget(self, nbytes):
        """Largest buffer of per-(layer, group) line records a call may allocate; beyond it the
        line list is walked in chunks (pb_lbl_set_record_budget)."""
        call('pb_lbl_set_record_budget', self._h, int(nbytes))

    @property
    def last_chunks(self):
        n = C.c_int(0)
        call('pb_lbl_last_chunks', self._h, C.byref(n))
        return n.value

    def set_concurrency(self, n):
        """The caller keeps n independent spectra in flight (pb_lbl_set_concurrency)."""
        call('pb_lbl_set_concurrency', self._h, int(n))

    @property
    def last_gather_kernel(self):
        m = C.c_int(0)
        call('pb_lbl_last_gather_mode', self._h, C.byref(m))
        base = {0: None, 1: 'k_ext_resample', 2: 'k_ext_staged', 3: 'k_ext_linterp',
                4: 'k_ext_scatter', 5: 'k_ext_rounds', 6: 'dynamic grids'}[m.value & 7]
        if m.value & 16:
            base = 'k_ext_wave+' + base
        return 'k_ext_resident+' + base if m.value & 8 else base

    def extinction(self, temp, dens, isoz, add=True, out=None, wbegin=0, wcount=None):
        """temp[L], dens[L,nmol], isoz[niso,L] device tensors -> ec[L,rows,wcount]."""
        nlayers = temp.shape[0]
        if wcount is None:
            wcount = self.nwave - wbegin
        rows = 1 if add else self.nrows_sep
        if out is None:
            alloc = torch.zeros if self.resolution else torch.empty
            out = alloc((nlayers, rows, wcount), dtype=torch.float64, device=temp.device)
        assert out.shape == (nlayers, rows, wcount) and out.is_contiguous()
        assert dens.shape == (nlayers, self.nmol) and isoz.shape == (self.niso, nlayers)
        call('pb_lbl_extinction', self._h, _ptr(out), int(wbegin), int(wcount), _ptr(temp),
             _ptr(dens), _ptr(isoz), isoz.stride(0), isoz.stride(1), nlayers, int(bool(add)),
             _stream())
        return out

    def extinction_begin(self, temp, dens, isoz, add=True, out=None, wbegin=0, wcount=None):
        """First half of extinction() for a wavenumber shard of a multi-GPU run: layer state +
        the records of the groups within reach of the shard, per-row maxima over those groups
        only.  All-reduce (MAX) kmax_tensor() over the ranks, then call extinction_end()."""
        nlayers = temp.shape[0]
        if wcount is None:
            wcount = self.nwave - wbegin
        rows = 1 if add else self.nrows_sep
        if out is None:
            alloc = torch.zeros if self.resolution else torch.empty
            out = alloc((nlayers, rows, wcount), dtype=torch.float64, device=temp.device)
        assert out.shape == (nlayers, rows, wcount) and out.is_contiguous()
        assert dens.shape == (nlayers, self.nmol) and isoz.shape == (self.niso, nlayers)
        call('pb_lbl_extinction_begin', self._h, _ptr(out), int(wbegin), int(wcount), _ptr(temp),
             _ptr(dens), _ptr(isoz), isoz.stride(0), isoz.stride(1), nlayers, int(bool(add)),
             _stream())
        return out

    def kmax_tensor(self):
        """The per-(layer, row) maxima of the plan as an int64 device tensor that ALIASES the
        library's buffer (bit patterns of non-negative doubles: integer MAX = double max)."""
        if getattr(self, '_kmax', None) is None:
            ptr, n = C.c_void_p(), C.c_int64(0)
            call('pb_lbl_kmax_buffer', self._h, C.byref(ptr), C.byref(n))

            class _Alias:
                __cuda_array_interface__ = {'shape': (n.value,), 'typestr': '<i8',
                                            'data': (ptr.value, False), 'version': 2}
            self._kmax = torch.as_tensor(_Alias(), device='cuda')
        return self._kmax

    def extinction_end(self):
        call('pb_lbl_extinction_end', self._h, _stream())

    def timing_begin(self, max_launches):
        call('pb_lbl_timing_begin', self._h, int(max_launches))

    def timing_end(self):
        """(summed gather-kernel milliseconds, launches) since timing_begin()."""
        ms, n = C.c_double(0), C.c_int(0)
        call('pb_lbl_timing_end', self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def last_work(self):
        """{fma_lanes_useful, fma_lanes_issued, live_records} of the last call, counted on the
        device (pb_lbl_last_work); None when that launch kept no packed records."""
        w = (C.c_int64 * 3)()
        call('pb_lbl_last_work', self._h, C.byref(w), _stream())
        if w[0] < 0:
            return None
        return dict(fma_lanes_useful=int(w[0]), fma_lanes_issued=int(w[1]),
                    live_records=int(w[2]))

    def last_table_samples(self):
        """Distinct Voigt-table samples the live records of the last call select (None when not
        counted): pb_lbl_last_table_samples."""
        n = C.c_int64(-1)
        call('pb_lbl_last_table_samples', self._h, C.byref(n), _stream())
        return None if n.value < 0 else int(n.value)

    def last_state(self, nlayers, rows):
        ofactor = np.zeros(nlayers, np.int32)
        kmax = np.zeros((nlayers, rows))
        call('pb_lbl_last_state', self._h, hptr(ofactor), hptr(kmax), nlayers, rows,
             _stream())
        return ofactor, kmax

    def last_layer_kinds(self, nlayers):
        """(resident[L] 0/1, block[L] doubles) of the last call: pb_lbl_last_layer_kinds."""
        resident = np.zeros(nlayers, np.int32)
        block = np.zeros(nlayers, np.int32)
        call('pb_lbl_last_layer_kinds', self._h, hptr(resident), hptr(block), nlayers, _stream())
        return resident, block

    def last_wave_layers(self, nlayers):
        """wave[L] 0/1: the layers of the last call the wave-autonomous kernel computed."""
        wave = np.zeros(nlayers, np.int32)
        call('pb_lbl_last_wave_layers', self._h, hptr(wave), nlayers, _stream())
        return wave

    def set_dyn_predict(self, on=True):
        """`resolution` plans, gather mode 'dynamic': plan every call from the last read-back of
        the layers' oversampling factors instead of synchronising the stream (pbhip.h:
        pb_lbl_set_dyn_predict); needed to capture such a call into a HIP graph."""
        call('pb_lbl_set_dyn_predict', self._h, int(bool(on)))

    def dyn_stats(self):
        """`resolution` plans, gather mode 'dynamic': (calls planned from the last read-back of the
        layers' factors -- no stream synchronisation --, synchronous calls, read-backs that
        contradicted the prediction their call was planned with)."""
        st = np.zeros(3, np.int64)
        call('pb_lbl_dyn_stats', self._h, hptr(st))
        return tuple(int(v) for v in st)

    def close(self):
        if self._h:
            call('pb_lbl_destroy', self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# Host pre-computes of the callers
# --------------------------------------------------------------------------
def transit_path(radius, nskip=0):
    """Chord segments between concentric shells for each impact parameter
    (pyratbay/atmosphere/atmosphere.py:737-802)."""
    rad = np.asarray(radius, float)[nskip:]
    # The reference squares SCALARS (`rad[i]**2`: libm pow), and pow(x, 2) is not always the
    # correctly rounded x*x that NumPy's array power computes (0.09 % of values differ by one
    # ulp).  Every square here is the scalar pow, like there: mixing the two forms makes
    # rad[r]**2 - rad[r]**2 non-zero, and its square root NaN, for one atmosphere in ~30.
    sq = np.array([x**2 for x in rad.tolist()], float)
    path = [np.empty(0) for _ in range(nskip)]
    for r in range(len(rad)):
        path.append(np.sqrt(sq[:r] - sq[r]) - np.sqrt(sq[1:r + 1] - sq[r]))
    return path


def pack_raypath(raypath, itop):
    """Lower triangle of transit_path(radius, itop) as one array (pbhip.h layout)."""
    rows = [np.asarray(p, float) for p in raypath[itop:]]
    for r, p in enumerate(rows):
        assert len(p) == r, 'raypath must come from transit_path(radius, itop)'
    return np.concatenate(rows) if rows else np.empty(0)


def transit_path_device(radius, itop=0):
    """atmosphere.transit_path on the device: radius[L] or [nw, L] (device) -> packed lower
    triangle [n(n-1)/2] or [nw, n(n-1)/2], n = L - itop (pb_transit_path)."""
    rad = radius if radius.dim() == 2 else radius.view(1, -1)
    nw, nlayers = rad.shape
    n = nlayers - itop
    out = torch.empty((nw, (n * (n - 1)) // 2), dtype=torch.float64, device=rad.device)
    call('pb_transit_path', _ptr(out), _ptr(rad.contiguous()), int(itop), nlayers, nw, _stream())
    return out if radius.dim() == 2 else out[0]


def interp_ec_batch(etable, ttable, temps, dens, out=None, tile_limit=None, row0=0, gate=None,
                    work=None):
    """interp_ec for a batch of walkers (assigning): temps[nw, L], dens[nw, L, S] ->
    ec[nw, L, W]; the table is read once per chunk of walkers.  tile_limit (int32[ceil(W/256)],
    device) / row0: only the layers a block of 256 columns can need are written
    (pb_interp_ec_batch_limited); gate (int32[1], device): the launch does nothing unless the
    flag is set, and runs on the `work` buffer its first pass filled."""
    nmol, ntemp, nlayers, nwave = etable.shape
    nw = temps.shape[0]
    assert temps.shape == (nw, nlayers) and dens.shape == (nw, nlayers, nmol)
    if out is None:
        out = torch.empty((nw, nlayers, nwave), dtype=torch.float64, device=etable.device)
    if work is None:
        work = torch.empty(nw * nlayers * 17 + 8, dtype=torch.float64, device=etable.device)
    if tile_limit is None and gate is None:
        call('pb_interp_ec_batch', _ptr(out), _ptr(etable), _ptr(ttable),
             _ptr(temps.contiguous()), _ptr(dens.contiguous()), _ptr(work), nmol, ntemp, nlayers,
             nwave, nw, _stream())
    else:
        call('pb_interp_ec_batch_limited', _ptr(out), _ptr(etable), _ptr(ttable),
             _ptr(temps.contiguous()), _ptr(dens.contiguous()), _ptr(work), nmol, ntemp, nlayers,
             nwave, nw, _ptr(tile_limit), int(row0), _ptr(gate), _stream())
    return out


def transit_spectrum_batch(ec, raypath, radius, rstar, itop, ibottom, maxdepth,
                           want_depth=False):
    """optical depth + transmission for a batch: ec[nw, L, W], raypath[nw, npath],
    radius[nw, L] -> spectrum[nw, W] (and depth[nw, L, W], ideep[nw, W] when asked for)."""
    nw, nlayers, nwave = ec.shape
    spectrum = torch.empty((nw, nwave), dtype=torch.float64, device=ec.device)
    depth = ideep = None
    if want_depth:
        depth = torch.empty_like(ec)
        ideep = torch.empty((nw, nwave), dtype=torch.int32, device=ec.device)
    nwork = _capi.lib().pb_transit_work_doubles(nlayers, int(itop), int(ibottom), nwave, nw)
    work = torch.empty(nwork, dtype=torch.float64, device=ec.device)
    call('pb_transit_spectrum_batch', _ptr(spectrum), _ptr(depth), _ptr(ideep), _ptr(ec),
         _ptr(raypath), _ptr(radius), float(rstar), int(itop), int(ibottom), float(maxdepth),
         nlayers, nwave, nw, _ptr(work), _stream())
    return (spectrum, depth, ideep) if want_depth else spectrum


def transit_spectrum_ordered(ec, raypath, radius, column, rstar, itop, ibottom, maxdepth,
                             tile_limit=None, flags=None, gate=None, out=None, work=None):
    """transit_spectrum_batch for ec[nw, L, W] whose columns are in the order `column` (int32[W]:
    grid index of each column): spectrum[nw, W] in GRID order.  Wavefronts stop at the row tile in
    which their 32 columns have all crossed maxdepth (pb_transit_spectrum_ordered).  With
    tile_limit / flags / gate: pb_transit_spectrum_limited (see interp_ec_batch)."""
    nw, nlayers, nwave = ec.shape
    spectrum = out if out is not None else torch.empty((nw, nwave), dtype=torch.float64,
                                                       device=ec.device)
    if work is None:
        nwork = _capi.lib().pb_transit_work_doubles(nlayers, int(itop), int(ibottom), nwave, nw)
        work = torch.empty(nwork, dtype=torch.float64, device=ec.device)
    if tile_limit is None and gate is None:
        call('pb_transit_spectrum_ordered', _ptr(spectrum), _ptr(ec), _ptr(raypath), _ptr(radius),
             _ptr(column), float(rstar), int(itop), int(ibottom), float(maxdepth), nlayers, nwave,
             nw, _ptr(work), _stream())
    else:
        call('pb_transit_spectrum_limited', _ptr(spectrum), _ptr(ec), _ptr(raypath), _ptr(radius),
             _ptr(column), float(rstar), int(itop), int(ibottom), float(maxdepth), nlayers, nwave,
             nw, _ptr(work), _ptr(tile_limit), _ptr(flags), _ptr(gate), _stream())
    return spectrum


def table_transit_supported(nmol, ntemp, nlayers, itop, ibottom, nwave):
    """Whether the one-pass form (table_transit_batch) exists for this shape -- never in the
    default library (an experiment: libpbhip_exp.so)."""
    if not _capi.experiments():
        return False
    return bool(_capi.lib().pb_table_transit_supported(int(nmol), int(ntemp), int(nlayers),
                                                       int(itop), int(ibottom), int(nwave)))


def table_transit_batch(etable, ttable, temps, dens, raypath, radius, rstar, itop, ibottom,
                        maxdepth):
    """interp_ec + optical depth + transmission of a batch of walkers in ONE pass
    (pb_table_transit_batch; experiments build of the library only): etable[S, T, L, W],
    temps[nw, L], dens[nw, L, S],
    raypath[nw, npath], radius[nw, L] -> spectrum[nw, W].  The interpolated extinction is
    the operand of the matrix products and is never stored."""
    nmol, ntemp, nlayers, nwave = etable.shape
    nw = temps.shape[0]
    assert temps.shape == (nw, nlayers) and dens.shape == (nw, nlayers, nmol)
    assert radius.shape == (nw, nlayers) and raypath.shape[0] == nw
    spectrum = torch.empty((nw, nwave), dtype=torch.float64, device=etable.device)
    if not _capi.experiments():
        call('pb_table_transit_batch')                     # (raises: not in libpbhip.so)
    nwork = _capi.lib().pb_table_transit_work_doubles(nmol, nlayers, int(itop), int(ibottom), nw)
    work = torch.empty(max(nwork, 8), dtype=torch.float64, device=etable.device)
    call('pb_table_transit_batch', _ptr(spectrum), _ptr(etable), _ptr(ttable),
         _ptr(temps.contiguous()), _ptr(dens.contiguous()), _ptr(raypath.contiguous()),
         _ptr(radius.contiguous()), float(rstar), int(itop), int(ibottom), float(maxdepth),
         nmol, ntemp, nlayers, nwave, nw, _ptr(work), _stream())
    return spectrum


def emission_flux_batch(ec, intervals, wn, temps, mu, weights, itop, ibottom, maxdepth,
                        column=None, tile_limit=None, flags=None, gate=None, out=None):
    """plane-parallel optical depth + emission flux for a batch: ec[nw, L, W],
    intervals[nw, L-1], temps[nw, L] -> flux[nw, W] (no cloud deck).  With `column` (int32[W]) the
    columns of ec and wn are in that order (grid index of each) and flux comes in grid order.
    tile_limit / flags / gate (ordered columns only): pb_emission_flux_limited, see
    interp_ec_batch."""
    nw, nlayers, nwave = ec.shape
    flux = out if out is not None else torch.empty((nw, nwave), dtype=torch.float64,
                                                   device=ec.device)
    if tile_limit is not None or gate is not None:
        call('pb_emission_flux_limited', _ptr(flux), _ptr(ec), _ptr(intervals.contiguous()),
             _ptr(wn), _ptr(temps.contiguous()), _ptr(mu), _ptr(weights), _ptr(column), len(mu),
             float(maxdepth), int(itop), int(ibottom), nlayers, nwave, nw, _ptr(tile_limit),
             _ptr(flags), _ptr(gate), _stream())
        return flux
    if column is not None:
        call('pb_emission_flux_ordered', _ptr(flux), _ptr(ec), _ptr(intervals.contiguous()),
             _ptr(wn), _ptr(temps.contiguous()), _ptr(mu), _ptr(weights), _ptr(column), len(mu),
             float(maxdepth), int(itop), int(ibottom), nlayers, nwave, nw, _stream())
        return flux
    call('pb_emission_flux_batch', _ptr(flux), _ptr(ec), _ptr(intervals.contiguous()), _ptr(wn),
         _ptr(temps.contiguous()), _ptr(mu), _ptr(weights), len(mu), float(maxdepth), int(itop),
         int(ibottom), nlayers, nwave, nw, _stream())
    return flux


# --------------------------------------------------------------------------
# Column stages (device tensors in, device tensors out)
# --------------------------------------------------------------------------
def optical_depth_transit(ec, raypath_packed, itop, ibottom, maxdepth):
    """optic_depth.py:103-112.  ec[L,W] -> depth[L,W], ideep[W] (int32)."""
    nlayers, nwave = ec.shape
    depth = torch.empty_like(ec)
    ideep = torch.empty(nwave, dtype=torch.int32, device=ec.device)
    call('pb_optical_depth_transit', _ptr(depth), _ptr(ideep), _ptr(ec),
         _ptr(raypath_packed), int(itop), int(ibottom), float(maxdepth), nlayers, nwave,
         _stream())
    return depth, ideep


def transit_spectrum(ec, raypath_packed, radius, rstar, itop, ibottom, maxdepth,
                     deck_rsurf=None, deck_itop=None, out=None):
    """optic_depth.py:103-112 + radiative_transfer.py:57-71 in one call:
    ec[L,W] -> spectrum[W], depth[L,W], ideep[W].  With an opaque cloud deck pass its
    radius and the index of the layer right below it (and ibottom = deck_itop + 1).
    out: a contiguous [W] tensor to receive the spectrum (a shard's slot of a gather buffer)."""
    nlayers, nwave = ec.shape
    depth = torch.empty_like(ec)
    ideep = torch.empty(nwave, dtype=torch.int32, device=ec.device)
    if out is not None:
        assert out.shape == (nwave,) and out.dtype == torch.float64 and out.is_contiguous()
    spectrum = out if out is not None else torch.empty(nwave, dtype=torch.float64,
                                                       device=ec.device)
    call('pb_transit_spectrum_deck', _ptr(spectrum), _ptr(depth), _ptr(ideep), _ptr(ec),
         _ptr(raypath_packed), _ptr(radius), float(rstar), int(itop), int(ibottom),
         float(maxdepth), -1 if deck_rsurf is None else int(deck_itop),
         0.0 if deck_rsurf is None else float(deck_rsurf), nlayers, nwave, _stream())
    return spectrum, depth, ideep


def patchy_transit_spectrum(ec, ec_cloud, fpatchy, raypath_packed, radius, rstar, itop,
                            maxdepth, deck_rsurf=None, deck_itop=None):
    """Patchy clouds, transit geometry (opacity/optic_depth.py:94-121 +
    pyrat/spectrum.py:350-363): the cloudy atmosphere is ec + ec_cloud (from itop down) with
    the opaque deck, if any, as its bottom; the clear one is ec over all layers; the spectrum
    is their fpatchy-weighted mean.  -> (spectrum, clear, cloudy), each [W]."""
    nlayers = ec.shape[0]
    ec_cloudy = ec.clone()
    ec_cloudy[itop:] += ec_cloud[itop:]
    ibottom = nlayers if deck_rsurf is None else int(deck_itop) + 1
    cloudy, _, _ = transit_spectrum(ec_cloudy, raypath_packed, radius, rstar, itop, ibottom,
                                    maxdepth, deck_rsurf, deck_itop)
    clear, _, _ = transit_spectrum(ec, raypath_packed, radius, rstar, itop, nlayers, maxdepth)
    return fpatchy * cloudy + (1.0 - fpatchy) * clear, clear, cloudy


def patchy_emission_flux(ec, ec_cloud, fpatchy, intervals, wn, temp, mu, weights, itop,
                         maxdepth, deck_tsurf=None, deck_itop=None):
    """Patchy clouds, plane-parallel emission (opacity/optic_depth.py:123-136 +
    pyrat/spectrum.py:366-385).  -> (flux, clear, cloudy), each [W]."""
    nlayers = ec.shape[0]
    ec_cloudy = ec.clone()
    ec_cloudy[itop:] += ec_cloud[itop:]
    ibottom = nlayers if deck_tsurf is None else int(deck_itop) + 1
    depth, ideep = plane_parallel_optical_depth(ec_cloudy, intervals, itop, ibottom, maxdepth)
    cloudy = emission_flux(depth, ideep, wn, temp, mu, weights, itop,
                           cloud_tsurf=deck_tsurf, cloud_itop=deck_itop)
    depth, ideep = plane_parallel_optical_depth(ec, intervals, itop, nlayers, maxdepth)
    # The reference's cloudy pass overwrites row deck_itop of its Planck array with the
    # cloud-top emission IN PLACE (spectrum/radiative_transfer.py:125-126) and the clear pass
    # then integrates that same array (pyrat/spectrum.py:380-383): its "clear" atmosphere
    # emits at the cloud-top temperature in that one layer.  Reproduced, not corrected.
    temp_clear = temp
    if deck_tsurf is not None:
        temp_clear = temp.clone()
        temp_clear[int(deck_itop)] = float(deck_tsurf)
    clear = emission_flux(depth, ideep, wn, temp_clear, mu, weights, itop)
    return fpatchy * cloudy + (1.0 - fpatchy) * clear, clear, cloudy


def plane_parallel_optical_depth(ec, intervals, itop, ibottom, maxdepth, depth=None):
    """optic_depth.py:122-130.  Rows below the stopping layer stay zero."""
    nlayers, nwave = ec.shape
    if depth is None:
        depth = torch.zeros_like(ec)
    ideep = torch.full((nwave,), nlayers - 1, dtype=torch.int32, device=ec.device)
    call('pb_plane_parallel_optical_depth', _ptr(depth), _ptr(ideep), _ptr(ec),
         _ptr(intervals), float(maxdepth), int(itop), int(ibottom), nlayers, nwave, _stream())
    return depth, ideep


def transmission(depth, ideep, radius, itop, rstar, deck_rsurf=None, deck_itop=None):
    """radiative_transfer.py:17-71 -> spectrum[W]; deck_rsurf / deck_itop = radius of an
    opaque cloud deck and index of the layer right below it."""
    nlayers, nwave = depth.shape
    spectrum = torch.empty(nwave, dtype=torch.float64, device=depth.device)
    call('pb_transmission_deck', _ptr(spectrum), _ptr(depth), _ptr(ideep), _ptr(radius),
         int(itop), float(rstar), -1 if deck_rsurf is None else int(deck_itop),
         0.0 if deck_rsurf is None else float(deck_rsurf), nlayers, nwave, _stream())
    return spectrum


def emission_flux(depth, ideep, wn, temp, mu, weights, rtop, want_intensity=False,
                  cloud_tsurf=None, cloud_itop=None):
    """pyrat/spectrum.py:366-377: Planck + intensity per mu + quadrature sum.  With an
    opaque cloud deck (radiative_transfer.py:121-131) the layer cloud_itop radiates at
    cloud_tsurf and is the deepest one seen."""
    nlayers, nwave = depth.shape
    flux = torch.empty(nwave, dtype=torch.float64, device=depth.device)
    inten = (torch.empty((len(mu), nwave), dtype=torch.float64, device=depth.device)
             if want_intensity else None)
    itop_cloud = -1
    if cloud_tsurf is not None:
        temp = temp.clone()
        temp[int(cloud_itop)] = float(cloud_tsurf)
        itop_cloud = int(cloud_itop)
    call('pb_emission_flux_deck', _ptr(flux), _ptr(inten), _ptr(depth), _ptr(ideep), _ptr(wn),
         _ptr(temp), _ptr(mu), _ptr(weights), len(mu), int(rtop), itop_cloud, nlayers, nwave,
         _stream())
    return (flux, inten) if want_intensity else flux


# the reference's rt_path values by family (constants/code_constants.py:83-102) -> (geometry of the
# radiative transfer here, what is made of the flux afterwards)
RT_PATHS = {
    'transit': ('transit', None),
    'emission': ('emission', 'emission'),
    'eclipse': ('emission', 'eclipse'),
    'f_lambda': ('emission', 'f_lambda'),
    'two_stream': ('two_stream', 'emission'),
    'emission_two_stream': ('two_stream', 'emission'),
    'eclipse_two_stream': ('two_stream', 'eclipse'),
}


def emission_observables(flux, kind='emission', starflux=None, rplanet=None, rstar=None,
                         f_dilution=None, wn=None, distance=None, in_place=False):
    """What the reference makes of a plane-parallel flux after the radiative transfer
    (pyrat/spectrum.py:394-405, eval()'s f_lambda conversion pyrat_obj.py:323-329), one launch:
    flux[W] -> (spectrum[W], fplanet[W]).
      fplanet = flux [* f_dilution]
      kind 'emission': spectrum = fplanet (the same tensor, as in the reference)
      kind 'eclipse' : spectrum = fplanet * (1/starflux * (rplanet/rstar)**2)
      kind 'f_lambda': spectrum = 10 * fplanet * (rplanet/distance * wn * 1e-4)**2
    in_place: fplanet is written over `flux`."""
    mode = {'emission': 0, 'eclipse': 1, 'f_lambda': 2}[kind]
    n = flux.shape[0]
    scale = 0.0
    if mode == 1:
        if starflux is None or rplanet is None or rstar is None:
            raise _capi.PbError('eclipse: starflux[W], rplanet and rstar are needed '
                                '(pyrat/argum.py:37-44)')
        assert starflux.shape == flux.shape
        scale = (float(rplanet) / float(rstar))**2
    if mode == 2:
        if wn is None or rplanet is None or distance is None:
            raise _capi.PbError('f_lambda: wn[W], rplanet and distance are needed')
        assert wn.shape == flux.shape
        scale = float(rplanet) / float(distance)
    if mode == 0 and f_dilution is None:
        return flux, flux                                   # (`spec.fplanet = spec.spectrum`)
    fplanet = flux if in_place else torch.empty_like(flux)
    spectrum = fplanet if mode == 0 else torch.empty_like(flux)
    call('pb_emission_observables', _ptr(spectrum), None if mode == 0 else _ptr(fplanet),
         _ptr(flux), _ptr(starflux), _ptr(wn), n, mode, 0 if f_dilution is None else 1,
         0.0 if f_dilution is None else float(f_dilution), scale, _stream())
    return spectrum, fplanet


def loglike(bandflux, data, uncert):
    """tools/retrieval_tools.py:98-104 for a batch of walkers: bandflux[nw, nbands] (or
    [nbands]) -> loglike[nw]; a non-finite value becomes -1e98, the reference's reject value."""
    bf = bandflux if bandflux.dim() == 2 else bandflux.view(1, -1)
    out = torch.empty(bf.shape[0], dtype=torch.float64, device=bf.device)
    call('pb_loglike', _ptr(out), _ptr(bf.contiguous()), _ptr(data), _ptr(uncert), bf.shape[0],
         bf.shape[1], _stream())
    return out


def internal_flux(wn, tint):
    """f_int of pyrat/spectrum.py:475-478 (Planck at tint scaled to sigma*tint^4)."""
    out = torch.empty(wn.shape[0], dtype=torch.float64, device=wn.device)
    call('pb_internal_flux', _ptr(out), _ptr(wn), float(tint), wn.shape[0], _stream())
    return out


def two_stream(depth, wn, temp, f_int=None, flux_top=None, rtop=0):
    """pyrat/spectrum.py:454-522 -> (flux_down, flux_up) [L,W]; the emission spectrum is
    flux_up[0].  flux_top = beta_irr*(rstar/smaxis)**2*starflux (or None)."""
    nlayers, nwave = depth.shape
    down = torch.empty((nlayers, nwave), dtype=torch.float64, device=depth.device)
    up = torch.empty((nlayers, nwave), dtype=torch.float64, device=depth.device)
    call('pb_two_stream', _ptr(down), _ptr(up), _ptr(depth), _ptr(wn), _ptr(temp),
         _ptr(f_int), _ptr(flux_top), int(rtop), nlayers, nwave, _stream())
    return down, up


def blackbody_wn_2D(wn, temp, last=None):
    B = torch.zeros((temp.shape[0], wn.shape[0]), dtype=torch.float64, device=wn.device)
    call('pb_blackbody_wn_2D', _ptr(B), _ptr(wn), wn.shape[0], _ptr(temp), temp.shape[0],
         _ptr(last), _stream())
    return B


def intensity(tau, ideep, planck, mu, rtop):
    nlayers, nwave = tau.shape
    out = torch.empty((mu.shape[0], nwave), dtype=torch.float64, device=tau.device)
    call('pb_intensity', _ptr(out), _ptr(tau), _ptr(ideep), _ptr(planck), _ptr(mu),
         mu.shape[0], int(rtop), nlayers, nwave, _stream())
    return out


def interp_ec(extinction, etable, ttable, temperatures, density, lay1, lay2, per_mol=False,
              assign=False):
    """_extcoeff.interp_ec[_per_mol]: accumulates into `extinction`; assign=True writes the
    rows lay1..lay2 instead (no need to zero them first)."""
    nmol, ntemp, nlayers, nwave = etable.shape
    call('pb_interp_ec_set' if assign else 'pb_interp_ec', _ptr(extinction), _ptr(etable),
         _ptr(ttable), _ptr(temperatures),
         _ptr(density), nmol, ntemp, nlayers, nwave, int(lay1), int(lay2), int(per_mol),
         _stream())
    return extinction


class PassBands:
    """A set of pass bands resident on the device (PassBand.set_sampling/integrate,
    pyratbay/spectrum/spec_tools.py:120-233).  Each band is (start index on the global
    wavenumber grid, response sampled on wn[start:start+count], height); for photon
    counting the caller folds the wavelength factor into the response."""

    def __init__(self, wn, bands):
        self.nbands = len(bands)
        self.wn = dev(wn)
        start = np.array([b[0] for b in bands], np.int32)
        count = np.array([len(b[1]) for b in bands], np.int32)
        offset = np.concatenate([[0], np.cumsum(count)[:-1]]).astype(np.int64)
        self.start, self.count = dev(start, torch.int32), dev(count, torch.int32)
        self.offset = dev(offset, torch.int64)
        self.response = dev(np.concatenate([np.asarray(b[1], float) for b in bands]))
        self.heights = dev(np.array([b[2] for b in bands], float))
        self.partial = torch.zeros(self.nbands, dtype=torch.float64, device='cuda')
        self.scale = None               # per-band factor after the heights (set_eclipse)

    def partial_integrate(self, spectrum_full, wbegin=0, wcount=None):
        """Un-scaled partial sums over the pairs whose left sample is in the shard."""
        if wcount is None:
            wcount = spectrum_full.shape[0] - wbegin
        call('pb_band_integrate', _ptr(self.partial), _ptr(spectrum_full), _ptr(self.wn),
             _ptr(self.start), _ptr(self.count), _ptr(self.response), _ptr(self.offset),
             self.nbands, int(wbegin), int(wcount), _stream())
        return self.partial

    def set_eclipse(self, rplanet, rstar, bandflux_star):
        """Eclipse geometry: integrate_batch() then returns band(fplanet) * rprs**2 / bandflux_star
        (Pyrat.band_integrate, pyrat_obj.py:662-665).  bandflux_star[nbands] = the band integrals
        of the stellar flux (pyrat/argum.py:86-90: `star_bandflux()` computes them here)."""
        rprs = float(rplanet) / float(rstar)
        self.scale = dev(rprs**2.0 / np.asarray(bandflux_star, float))
        return self

    def star_bandflux(self, starflux):
        """bandflux_star of pyrat/argum.py:86-90: the bands' integrals of starflux[W] -> [nbands]
        (host array)."""
        scale, self.scale = self.scale, None
        try:
            out = self.integrate_batch(dev(starflux).view(1, -1))[0]
        finally:
            self.scale = scale
        return out.cpu().numpy()

    def integrate_batch(self, spectra, out=None, f_dilution=None):
        """Band fluxes (heights applied; then the walkers' dilution factors f_dilution[nw], if
        given, and the eclipse factor, if set) of full-grid spectra[nw, W] -> [nw, nbands]."""
        nw, nwave = spectra.shape
        if out is None:
            out = torch.empty((nw, self.nbands), dtype=torch.float64, device=spectra.device)
        call('pb_band_integrate_batch', _ptr(out), _ptr(spectra), _ptr(self.wn),
             _ptr(self.start), _ptr(self.count), _ptr(self.response), _ptr(self.offset),
             _ptr(self.heights), self.nbands, nwave, nw, _stream())
        if self.scale is not None or f_dilution is not None:
            assert f_dilution is None or f_dilution.shape == (nw,)
            call('pb_band_scale', _ptr(out), _ptr(self.scale), _ptr(f_dilution), self.nbands, nw,
                 _stream())
        return out


def default_quadrature():
    """(mu, weights) of the reference when `quadrature` is unset: raygrid = 0, 20, 40, 60, 80
    degrees, weights = the solid angle between the mid-points (pyrat/spectrum.py:30-58)."""
    raygrid = np.radians([0.0, 20.0, 40.0, 60.0, 80.0])
    bounds = np.linspace(0, 0.5 * np.pi, len(raygrid) + 1)
    bounds[1:-1] = 0.5 * (raygrid[:-1] + raygrid[1:])
    return np.cos(raygrid), np.pi * (np.sin(bounds[1:])**2 - np.sin(bounds[:-1])**2)


def _legendre_newton(n):
    """Gauss-Legendre nodes and weights on [-1, 1] by Newton's iteration on P_n in extended
    precision, ascending nodes.  Correctly rounded to ~1 ulp -- which is NOT what the reference
    uses: SciPy's roots_legendre (Golub-Welsch eigenvalues + one Newton step) is up to 3 ulp off
    in the nodes and up to 1.5e-13 relative in the weights at n <= 16 (tests/test_host_logic.py)."""
    ld = np.longdouble
    k = np.arange(1, n + 1, dtype=ld)
    x = np.cos(np.pi * (k - ld(0.25)) / (n + ld(0.5)))

    def pn(x):
        p0, p1 = np.ones_like(x), x.copy()
        for j in range(2, n + 1):
            p0, p1 = p1, ((2 * j - 1) * x * p1 - (j - 1) * p0) / j
        return p1, n * (x * p1 - p0) / (x * x - 1)
    for _ in range(60):
        p, dp = pn(x)
        dx = p / dp
        x = x - dx
        if np.max(np.abs(dx)) < 1e-19:
            break
    _, dp = pn(x)
    w = 2 / ((1 - x * x) * dp * dp)
    return x[::-1].astype(float), w[::-1].astype(float)


def gauss_quadrature(n, use_scipy=True):
    """(mu, weights) of the reference for `quadrature = n` (pyrat/spectrum.py:41-49):
    Gauss-Legendre nodes x_i, weights w_i of order n mapped to q = (x + 1) / 2, mu = sqrt(q),
    weights = pi/2 w -- the flux integral  2 pi Int_0^1 I(mu) mu dmu = pi Int_0^1 I dq.  The
    reference takes (x, w) from scipy.special.p_roots; so does this function when SciPy is
    importable (same call of the same third-party library: bit-identical mu and weights --
    tests/golden/g18_p_roots.npz holds SciPy 1.15.3's values); without SciPy (or with
    use_scipy=False) the nodes come from _legendre_newton (within 3 ulp / 1.5e-13 of SciPy's).
    n <= 16: the emission kernels keep at most 16 running sums per column."""
    n = int(n)
    if not 1 <= n <= 16:
        raise ValueError(f'quadrature = {n}: 1 ... 16 nodes are supported')
    nodes = weights = None
    if use_scipy:
        try:
            from scipy.special import roots_legendre
            nodes, weights = roots_legendre(n)
        except ImportError:
            pass
    if nodes is None:
        nodes, weights = _legendre_newton(n)
    qnodes = 0.5 * (nodes + 1.0)
    return np.sqrt(qnodes), 0.5 * np.pi * weights


# --------------------------------------------------------------------------
# Whole-path model: the three timed stages of Pyrat.run() (pyrat_obj.py:203-214)
# --------------------------------------------------------------------------
class LBLSpectrum:
    """extinction -> optical depth -> spectrum for one wavenumber shard, all on device.

    `case` is a dict as produced by pyratbay_amd.synth.lbl_case (or assembled by a caller
    from a real Pyrat object: the same arrays the reference hands to its C extensions).
    """

    def __init__(self, case, rt_path='transit', wbegin=0, wcount=None, itop=0,
                 quadrature_mu=None, quadrature_weights=None, keep_flat=False,
                 voigt=None, lines=None, tint=0.0, flux_top=None, continuum=None,
                 continuum_density=None, timestamps=True, materialize_depth=True,
                 predict_runs=False, starflux=None, rplanet=None, f_dilution=None,
                 distance=None):
        require_gpu()
        # rt_path: any of the reference's (constants/code_constants.py:83-102) or 'two_stream'
        # (= emission_two_stream).  self.rt_path is the GEOMETRY of the radiative transfer
        # ('transit', 'emission', 'two_stream'); self.observable what is made of an emission-type
        # flux afterwards ('emission', 'eclipse', 'f_lambda'; pyrat/spectrum.py:394-405) from
        # starflux[nwave], rplanet (with atm['rstar']), f_dilution, distance.
        if rt_path not in RT_PATHS:
            raise _capi.PbError(f'rt_path {rt_path!r}: select from {sorted(RT_PATHS)}')
        self.rt_path_name = rt_path
        rt_path, self.observable = RT_PATHS[rt_path]
        # per-stage HIP-event timers behind the `timestamps` property (the reference's
        # pyrat.timestamps keys); timestamps=False: run() records no events
        self._timer = StageTimer() if timestamps else None
        # materialize_depth=False (transit geometry): run() computes the optical depths, applies
        # the reference's exit rule and integrates the spectrum in ONE pass on the matrix cores
        # (pb_transit_spectrum_batch with one "walker": tau = Q . ec) without
        # writing depth[L, W] / ideep[W] -- `self.depth` and `self.ideep` then stay None.  The
        # default keeps the reference's outputs (pyrat.od.depth, pyrat.od.ideep).
        self.materialize_depth = bool(materialize_depth)
        g, atm, ln, iso, vg = (case['grid'], case['atm'], case['lines'], case['iso'],
                               case['voigt'])
        self.case = case
        self.rt_path = rt_path
        # optional continuum terms (pyratbay_amd.continuum.Continuum on this shard's grid)
        # and the host-side number densities {species: n[L]} they use
        self.continuum, self.continuum_density = continuum, continuum_density
        self.nwave = g['nwave']
        self.nlayers = atm['nlayers']
        self.wbegin = wbegin
        self.wcount = self.nwave - wbegin if wcount is None else wcount
        self.itop = itop
        self.maxdepth = case['maxdepth']
        # (resolution mode reads the reference layout only: keep_flat = 2 keeps no second copy)
        interpolate = g.get('resolution') is not None or bool(g.get('interpolate'))
        if interpolate and not keep_flat:
            keep_flat = 2
        self.voigt = voigt or VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'],
                                               g['ownstep'], g['wnosamp'], keep_flat)
        self.lines = lines or LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'],
                                       len(iso['isomass']), g['own'])
        # a constant-resolving-power (or constant-wavelength-step) output grid: the kept samples
        # are interpolated from the dynamic grid (_extcoeff.c:320-326) and ACCUMULATED into ec
        self.resolution = interpolate
        self.lbl = LBL(self.voigt, self.lines, g['wn'], g['divisors'], atm['mol_radius'],
                       atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                       iso['isoiext'], vg['cutoff'], case['ethresh'],
                       resolution=self.resolution, max_layers=self.nlayers)
        if self.resolution:
            # an object made for many spectra: the one-time constant-step sub-plans of the layers'
            # dynamic grids pay off from the second spectrum on (a bare LBL plan keeps the direct gather)
            self.lbl.set_gather_mode('dynamic')
            # predict_runs: plan every call from the last read-back of the layers' oversampling
            # factors instead of synchronising the stream in every call (LBL.set_dyn_predict;
            # results to 1e-12 of the default, bit for bit for a steady atmosphere; capturable)
            if predict_runs:
                self.lbl.set_dyn_predict(True)
        self.predict_runs = bool(predict_runs) and self.resolution
        # atmosphere state, resident (+ the host copy of the temperatures that the continuum
        # terms take their per-layer factors from)
        self.temp_host = np.array(atm['temp'], float)
        self.temp = dev(atm['temp'])
        self.dens = dev(atm['dens'])
        self.isoz = dev(iso['isoz'])
        self.radius = dev(atm['radius'])
        self.rstar = float(atm['rstar'])
        self.wn = dev(g['wn'][wbegin:wbegin + self.wcount])
        if rt_path == 'transit':
            self.raypath = dev(pack_raypath(transit_path(atm['radius'], itop), itop))
        else:
            self.intervals = dev(-np.diff(atm['radius']))
            if quadrature_mu is None:
                quadrature_mu, quadrature_weights = default_quadrature()
            self.mu = dev(quadrature_mu)
            self.weights = dev(quadrature_weights)
        if rt_path == 'two_stream':
            # rt_path emission_two_stream: depth without the maxdepth stop
            # (opacity/optic_depth.py:124-125), internal flux, optional irradiation
            self.maxdepth = np.inf
            self.f_int = internal_flux(self.wn, tint)
            self.flux_top = None if flux_top is None else dev(
                np.asarray(flux_top)[wbegin:wbegin + self.wcount])
            self.flux_down = self.flux_up = None
        self.ec = torch.empty((self.nlayers, 1, self.wcount), dtype=torch.float64,
                              device='cuda')
        self.depth = self.ideep = self.spectrum = None
        # emission-type paths: the planet's flux (after f_dilution) beside `spectrum`, as the
        # reference's spec.fplanet; eclipse: spectrum = fplanet / starflux * (rplanet/rstar)^2
        self.fplanet = None
        self.f_dilution = f_dilution
        self.rplanet = rplanet if rplanet is not None else atm.get('rplanet')
        self.distance = distance
        self.starflux = None
        if self.observable == 'eclipse':
            if starflux is None or self.rplanet is None:
                raise _capi.PbError(f'rt_path {self.rt_path_name!r} needs starflux[nwave] and '
                                    'rplanet (pyrat/argum.py:37-44)')
            self.starflux = dev(np.asarray(starflux, float)[wbegin:wbegin + self.wcount])
        # the TLI file's partition-function tables (from_tli): set_atmosphere() without `isoz`
        # evaluates them at the new temperatures on the device
        self.partition = None
        # a [wcount] tensor the transit spectrum is written to instead of a fresh one (the shard's
        # slot of a gather buffer: dist.SpectrumGather(uniform=True))
        self.spectrum_out = None
        # set to a function(tensor) that all-reduces (MAX) over the ranks to switch the
        # extinction of a wavenumber shard to its two-phase form (dist.kmax_allreduce)
        self.kmax_exchange = None

    @classmethod
    def from_tli(cls, tlifiles, atm, grid, *, nlor=100, ndop=50, extent=300.0, cutoff=25.0,
                 dlratio=0.1, lorentz=None, doppler=None, tmin=100.0, tmax=3000.0,
                 ethresh=1e-30, maxdepth=10.0, skip_species=(), iso_numbering='file', **kw):
        """TLI file(s) + atmosphere + spectral grid -> a model ready to run(): what
        Line_By_Line.__init__ / Voigt.__init__ assemble before the reference's first extinction
        call (pyratbay/pyrat/line_by_line.py:120-200, pyrat/voigt.py:20-149), with nothing taken
        from a fixture -- lines and isotope data from the file(s) (tli.read_tli on the grid's
        range, databases concatenated in file order), isotope -> species indices by molecule
        name, Z_i(T_layer) by tli's restatement of the reference's interp1d, Voigt width grids
        from the atmosphere (or given: the reference's voigt_dmin/dmax/lmin/lmax keys).

        iso_numbering: how the lines of a file with SEVERAL databases find their isotope.  The
        file stores each line's isotope index relative to its own database (lread.py:181-209,
        309).  'file' (default): numbered over the file's databases, i.e. every line gets its own
        isotope.  'reference': as Line_By_Line does (line_by_line.py:114-119: the stored index +
        the isotope count of the previous FILES) -- in a multi-database file the lines of the
        second database then use the first database's isotope data, and the list steps back in
        wavenumber within an isotope id, which pb_lines_create refuses (the reference's result
        on such a list depends on its one-way Doppler-index search; fixture G16, run `onefile`,
        is pinned by the CPU-side checker only).  One database per file, the layout of the reference's
        own configurations, is the same either way.

        atm: dict with temp[L], dens[L, nspecies] (cm-3), radius[L], press[L] (bar; for the width
        grids), species (names), mol_mass, mol_radius (cm), rstar.  grid: synth.spectral_grid /
        resolution_grid / wlstep_grid (wn, own, ownstep, onwave, wnosamp, divisors, wnlow,
        wnhigh)."""
        from . import synth, tli
        paths = [tlifiles] if isinstance(tlifiles, (str, bytes, os.PathLike)) else list(tlifiles)
        species = list(atm['species'])
        dbs, lwn, gf, elow, lid = [], [], [], [], []
        niso = 0
        # the reference selects the lines of [spec.wnlow, spec.wnhigh] (pyrat/opacity.py:46-47,
        # 105): the CONFIGURED boundaries -- wnhigh can lie up to one step above wn[-1], and a line
        # in between still throws its wing onto the grid
        wn_lo = float(grid.get('wnlow', grid['wn'][0]))
        wn_hi = float(grid.get('wnhigh', grid['wn'][-1]))
        for path in paths:
            d, wn_, gf_, el_, stored, meta = tli.read_tli(path, wn_lo, wn_hi)
            dbs += d
            lwn.append(wn_); gf.append(gf_); elow.append(el_)
            if iso_numbering == 'reference':
                lid.append(stored.astype(np.int32) + niso)
            elif iso_numbering == 'file':
                lid.append(meta['iso_global'].astype(np.int32) + niso)
            else:
                raise ValueError("iso_numbering: 'file' or 'reference'")
            niso += sum(len(db['isotopes']) for db in d)
        isoimol, isomass, isoratio = [], [], []
        for db in dbs:
            if db['molecule'] not in species:
                raise ValueError(f"The species '{db['molecule']}' is not present in the "
                                 'atmosphere, required for LBL calculation')
            isoimol += [species.index(db['molecule'])] * len(db['isotopes'])
            isomass += list(db['iso_mass'])
            isoratio += list(db['iso_ratio'])
        isoimol = np.asarray(isoimol, np.int32)
        # rows of the un-added extinction: one per line-carrying species, in np.unique's order
        # (line_by_line.py:177-188); skip_species flags their isotopes -1 (extinction.py:165-168)
        carriers = sorted({species[i] for i in isoimol})
        isoiext = np.asarray([carriers.index(species[i]) for i in isoimol], np.int32)
        for name in skip_species:
            if name in carriers:
                isoiext[isoiext == carriers.index(name)] = -1
        iso = dict(isoimol=isoimol, isomass=np.asarray(isomass, float),
                   isoratio=np.asarray(isoratio, float), isoiext=isoiext,
                   isoz=tli.iso_partition(dbs, atm['temp']))
        if lorentz is None or doppler is None:
            used = np.unique(isoimol)
            lor, dop = synth.voigt_widths(grid['wn'], atm['press'],
                                          np.asarray(atm['mol_mass'])[used],
                                          np.asarray(atm['mol_radius'])[used], nlor, ndop,
                                          tmin, tmax)
            lorentz = lor if lorentz is None else lorentz
            doppler = dop if doppler is None else doppler
        lorentz, doppler = np.asarray(lorentz, float), np.asarray(doppler, float)
        size = synth.voigt_sizes(lorentz, doppler, extent, cutoff, grid['ownstep'],
                                 grid['onwave'], dlratio)
        atm = dict(atm)
        atm['nlayers'] = len(atm['temp'])
        case = dict(grid=grid, atm=atm, iso=iso,
                    lines=dict(lwn=np.concatenate(lwn), elow=np.concatenate(elow),
                               gf=np.concatenate(gf), lid=np.concatenate(lid)),
                    voigt=dict(lorentz=lorentz, doppler=doppler, size=size, extent=extent,
                               cutoff=cutoff, dlratio=dlratio),
                    ethresh=ethresh, maxdepth=maxdepth)
        model = cls(case, **kw)
        model.partition = PartitionTable(dbs)
        model.databases = dbs
        return model

    def set_atmosphere(self, temp, dens, isoz=None, radius=None, continuum_density=None):
        """New temperature / number-density / partition-function (and radius) profiles for the
        next run().  isoz=None (models made by from_tli): Z_i(T) is interpolated from the file's
        tables at the new temperatures on the device, as the reference does on every extinction
        call (line_by_line.py:219-222); a temperature outside a table raises ValueError.  With a
        Continuum attached pass its number densities {species: n[L]} too: every opacity term of
        the next run then sees the SAME atmosphere."""
        self.temp_host = np.array(temp.cpu().numpy() if isinstance(temp, torch.Tensor) else temp,
                                  float)
        self.temp.copy_(dev(temp))
        self.dens.copy_(dev(dens))
        if isoz is not None:
            self.isoz.copy_(dev(isoz))
        elif self.partition is not None:
            self.partition.evaluate(self.temp, out=self.isoz)
        else:
            raise _capi.PbError('set_atmosphere: pass isoz[niso, L] (this model has no '
                                'partition-function tables: it was not made by from_tli)')
        if continuum_density is not None:
            self.continuum_density = continuum_density
        elif self.continuum is not None:
            raise _capi.PbError('set_atmosphere: this model has continuum terms, pass their '
                                'number densities (continuum_density) with the new atmosphere')
        if radius is not None:
            self.radius.copy_(dev(radius))
            if self.rt_path == 'transit':
                self.raypath.copy_(dev(pack_raypath(transit_path(radius, self.itop),
                                                    self.itop)))
            else:
                self.intervals.copy_(dev(-np.diff(radius)))

    def extinction(self):
        if self.resolution:
            self.ec.zero_()              # the interpolating kernel adds to what it finds
        if self.kmax_exchange is not None and not self.resolution:
            # (`resolution` mode: the dynamic-grid path takes the maxima over every line itself
            # and a shard equals the slice of the whole call bit for bit -- no exchange)
            # wavenumber shard of a multi-GPU run: every rank derives the records (and the
            # strengths, the exp() work) of its own groups only; the per-row maxima that set
            # the ethresh threshold are made global by ONE small all-reduce(MAX)
            self.lbl.extinction_begin(self.temp, self.dens, self.isoz, add=True, out=self.ec,
                                      wbegin=self.wbegin, wcount=self.wcount)
            self.kmax_exchange(self.lbl.kmax_tensor())
            self.lbl.extinction_end()
        else:
            self.lbl.extinction(self.temp, self.dens, self.isoz, add=True, out=self.ec,
                                wbegin=self.wbegin, wcount=self.wcount)
        if self.continuum is not None:
            self.continuum.add(self.ec.view(self.nlayers, self.wcount), self.temp_host,
                               self.continuum_density)
        return self.ec

    def optical_depth(self):
        ec = self.ec.view(self.nlayers, self.wcount)
        if self.rt_path == 'transit':
            self.depth, self.ideep = optical_depth_transit(
                ec, self.raypath, self.itop, self.nlayers, self.maxdepth)
        else:
            self.depth, self.ideep = plane_parallel_optical_depth(
                ec, self.intervals, self.itop, self.nlayers, self.maxdepth)
        return self.depth, self.ideep

    def rt(self):
        if self.rt_path == 'transit':
            self.spectrum = transmission(self.depth, self.ideep, self.radius, self.itop,
                                         self.rstar)
        elif self.rt_path == 'two_stream':
            self.flux_down, self.flux_up = two_stream(self.depth, self.wn, self.temp,
                                                      self.f_int, self.flux_top, self.itop)
            self.spectrum = self.flux_up[0]
        else:
            self.spectrum = emission_flux(self.depth, self.ideep, self.wn, self.temp,
                                          self.mu, self.weights, self.itop)
        if self.rt_path != 'transit':
            # f_dilution, eclipse ratio (pyrat/spectrum.py:394-405); 'f_lambda' stays in
            # erg s-1 cm-2 cm here as in the reference's run(): observed() converts
            kind = 'eclipse' if self.observable == 'eclipse' else 'emission'
            self.spectrum, self.fplanet = emission_observables(
                self.spectrum, kind, self.starflux, self.rplanet, self.rstar, self.f_dilution,
                in_place=self.rt_path != 'two_stream')    # (flux_up[0] stays what two_stream made)
        return self.spectrum

    def get_ec(self, layer):
        """Pyrat.get_ec(layer) for the line-by-line model (pyrat_obj.py:700-719 ->
        line_by_line.py:224-230): the cross sections of ONE layer per species (`add = 0`,
        extinction.py:155-158) times that species' number density -> (ec[nspec, wcount] in cm-1 on
        the device, labels).  A species is a row of `isoiext`; its label is atm['species'] of the
        molecule its isotopes belong to (the row index when the case names none).  (The reference
        multiplies every row by `density[layer]` of ALL the model's species at once, which only
        broadcasts for a single-species model; here each row takes its own species' density.)"""
        atm, iso = self.case['atm'], self.case['iso']
        layer = int(layer)
        if not 0 <= layer < self.nlayers:
            raise _capi.PbError(f'get_ec: layer {layer} outside 0 ... {self.nlayers - 1}')
        sl = slice(layer, layer + 1)
        ec = self.lbl.extinction(self.temp[sl], self.dens[sl].contiguous(),
                                 self.isoz[:, sl].contiguous(), add=False, wbegin=self.wbegin,
                                 wcount=self.wcount)[0]
        isoiext = np.asarray(iso['isoiext'])
        isoimol = np.asarray(iso['isoimol'])
        imol = [int(isoimol[np.flatnonzero(isoiext == r)[0]]) if np.any(isoiext == r) else -1
                for r in range(ec.shape[0])]
        dens = self.dens[layer]
        scale = torch.stack([dens[m] if m >= 0 else torch.zeros_like(dens[0]) for m in imol])
        names = atm.get('species')
        labels = [str(names[m]) if names is not None and m >= 0 else str(r)
                  for r, m in enumerate(imol)]
        # ec[r, :] *= scale[r] (pb_band_scale with the rows as its 'walkers')
        call('pb_band_scale', _ptr(ec), None, _ptr(scale.contiguous()), ec.shape[1], ec.shape[0],
             _stream())
        return ec, labels

    def observed(self):
        """The last spectrum as eval() returns it (pyrat_obj.py:323-329): rt_path 'f_lambda'
        converts the planet's flux to W m-2 um-1 at `distance`; every other path: `spectrum`."""
        if self.observable != 'f_lambda':
            return self.spectrum
        return emission_observables(self.fplanet, 'f_lambda', rplanet=self.rplanet, wn=self.wn,
                                    distance=self.distance)[0]

    def capture(self):
        """Capture one run() into a HIP graph (torch.cuda.CUDAGraph on a side stream) and
        return a replay function: the whole step -- layer state, records, gather, optical
        depth, spectrum -- is then ONE graph launch, with inputs read from and outputs
        written to the same device buffers (update the atmosphere with set_atmosphere()).
        The first call allocates workspaces, so it runs once eagerly before the capture.
        The `resolution` mode's dynamic-grid path needs predict_runs=True: its launches depend on
        the layers' oversampling factors, which the default form reads back in every call; the
        captured plan is the one of the atmosphere at capture time, and layers of a later
        atmosphere that it does not fit are computed by the direct gather inside the graph."""
        if self.resolution and self.lbl.gather_mode == 'dynamic' and not self.predict_runs:
            raise RuntimeError("capture(): the dynamic-grid path of the `resolution` mode reads "
                               "the layers' factors back on every call and cannot be captured; "
                               "LBLSpectrum(..., predict_runs=True) plans its calls from the last "
                               "read-back instead, lbl.set_gather_mode('auto') selects the direct "
                               "gather")
        self.run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.run()                         # warm-up on the capture stream
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                out = self.run()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = graph

        def replay():
            graph.replay()
            return out
        return replay

    def run(self):
        """One spectrum: the 'extinction', 'odepth' and 'spectrum' stages (the last two
        in one library call for the transit geometry, which marks their boundary itself)."""
        t = self._timer
        if t is not None:
            t.start('extinction')
        self.extinction()
        if t is not None:
            t.mark('extinction', 'odepth')
        if self.rt_path == 'transit' and not self.materialize_depth:
            if t is not None:
                t.mark('odepth', 'spectrum')        # (no separate stage: counted under 'spectrum')
            self.depth = self.ideep = None
            self.spectrum = transit_spectrum_batch(
                self.ec.view(1, self.nlayers, self.wcount), self.raypath.view(1, -1),
                self.radius.view(1, -1), self.rstar, self.itop, self.nlayers, self.maxdepth)[0]
            if t is not None:
                t.mark('spectrum')
            return self.spectrum
        if self.rt_path == 'transit':
            self.spectrum, self.depth, self.ideep = transit_spectrum(
                self.ec.view(self.nlayers, self.wcount), self.raypath, self.radius,
                self.rstar, self.itop, self.nlayers, self.maxdepth, out=self.spectrum_out)
            if t is not None:
                t.mark('spectrum')
            return self.spectrum
        self.optical_depth()
        if t is not None:
            t.mark('odepth', 'spectrum')
        out = self.rt()
        if t is not None:
            t.mark('spectrum')
        return out

    @property
    def timestamps(self):
        """Seconds of the last run() by stage, with the reference's keys 'extinction',
        'odepth', 'spectrum' (pyrat_obj.py:203-214).  Waits for that run to finish."""
        if self._timer is None:
            raise _capi.PbError('this model was built with timestamps=False')
        return self._timer.read()


class SpectrumPipeline:
    """Consecutive, independent spectra of one line-by-line model kept in flight on `depth` HIP
    streams (default 2): spectrum i+1 starts while spectrum i is still finishing.

    Why: one spectrum of C2 is a chain of launches whose dominant one, the extinction gather,
    runs ~2000 workgroups of 0.3-0.7 ms on 1024 slots -- its last fifth is a tail in which most
    of the chip idles (measured: 79 % of the slot-time busy), and the small launches around it
    (layer state, records, ray paths, depth, spectrum) cannot fill a chip either.  A second
    spectrum on another stream fills those holes: +19 % spectra/s at C2 on one MI355X.  The
    callers this serves compute many independent spectra anyway: the temperature loop of
    `compute_opacity` (pyrat/extinction.py:100-122), the walkers of a retrieval.

    Every context has its own plan (records, per-layer state, ec, depth) and shares the Voigt
    table and the line list, which a run only reads.  A context's output buffers are reused by
    its next submit(): consume (or copy) a result before submitting `depth` more spectra.
    Results are bit-identical to LBLSpectrum.run() of the same atmosphere
    (tests/test_gpu_pipeline.py::test_spectrum_pipeline_equals_serial_runs)."""

    def __init__(self, case, depth=2, **kw):
        require_gpu()
        first = LBLSpectrum(case, **kw)
        kw = dict(kw, voigt=first.voigt, lines=first.lines)
        self.models = [first] + [LBLSpectrum(case, **kw) for _ in range(depth - 1)]
        for m in self.models:
            m.lbl.set_concurrency(depth)
        self.streams = side_streams(depth)
        self.done = [None] * depth          # completion event of each context's last spectrum
        self.count = 0

    @property
    def depth(self):
        return len(self.models)

    def submit(self, atmosphere=None):
        """Enqueue one spectrum (optionally of a new atmosphere: the arguments of
        LBLSpectrum.set_atmosphere as a tuple or dict) and return (spectrum, event): the
        device tensor is complete once `event` has fired (flush() waits for all of them)."""
        j = self.count % len(self.models)
        self.count += 1
        model, stream = self.models[j], self.streams[j]
        caller = torch.cuda.current_stream()
        if not caller.query():                              # (an idle stream has nothing to wait for)
            stream.wait_stream(caller)                      # inputs made on the caller's stream
        with torch.cuda.stream(stream):
            if isinstance(atmosphere, dict):
                model.set_atmosphere(**atmosphere)
            elif atmosphere is not None:
                model.set_atmosphere(*atmosphere)
            out = model.run()
            event = torch.cuda.Event()
            event.record(stream)
        # the result was allocated on the side stream and will be read on the caller's: keep the
        # caching allocator from handing its memory out again before the caller's reads are done
        out.record_stream(caller)
        self.done[j] = event
        return out, event

    def flush(self):
        """Make the caller's stream wait for every spectrum submitted so far."""
        cur = torch.cuda.current_stream()
        for event in self.done:
            if event is not None:
                cur.wait_event(event)


class TableSpectrum:
    """Retrieval inner loop on sampled cross sections (Line_Sample path,
    pyratbay/opacity/line_sampling.py:394-463 -> _extcoeff.interp_ec): the table
    etable[nspec, ntemp, nlayers, nwave] stays resident; each eval() interpolates it to the
    layer temperatures, weights by the species densities, and runs optical depth + RT.

    column_order='auto' (default) has a ONE-TIME cost in the first eval_bands() call of a model
    with >= 64 columns: the first walker's temperatures are checked on the host (one stream
    synchronisation), its spectrum is computed to order the columns by optical depth
    (order_columns: interpolation + transit/plane-parallel depth + a sort) and a permuted SECOND
    COPY of the table is made (2 x the table's memory from then on; skipped when
    1.25 x table + 2 x the batch's ec buffer do not fit in free memory, when the ordered kernels
    do not support the shape -- transit geometry with more than 128 impact parameters -- or while
    the stream is being captured into a graph).  Every later call is launch-only.  To keep the first call
    free of both, call order_columns(temp, dens) yourself during set-up or pass
    column_order=None (grid order).  In a multi-rank run every rank orders by its own first
    walker: results do not depend on the order, memory use per rank is the same 2 x table."""

    def __init__(self, etable, ttable, wn, radius, rstar, rt_path='transit', itop=0,
                 maxdepth=10.0, quadrature_mu=None, quadrature_weights=None, continuum=None,
                 timestamps=True, column_order='auto'):
        require_gpu()
        self._timer = StageTimer() if timestamps else None
        # eval_bands: the order the columns are worked in (see order_columns).
        # 'auto': taken from the first walker of the first batch; None: grid order
        self.column_order = None
        self.etable_ordered = None
        # transit geometry, columns ordered by order_columns(): last row tile a block of 256
        # ordered columns can need (int32, device) -- see _eval_chunk; None: every layer
        self.tile_limit = None
        self.tile_margin = int(os.environ.get('PB_C5_MARGIN', '4'))
        if isinstance(column_order, str) and column_order == 'auto' and \
                os.environ.get('PB_COLUMN_ORDER', '1') == '0':
            column_order = None                       # (A/B switch: grid order)
        self._auto_order = isinstance(column_order, str) and column_order == 'auto'
        if isinstance(column_order, str) and not self._auto_order:
            raise ValueError("column_order: 'auto', None or a permutation of range(nwave)")
        if column_order is not None and not self._auto_order:
            self._pending_order = column_order
        else:
            self._pending_order = None
        self.continuum = continuum          # pyratbay_amd.continuum.Continuum or None
        self.etable = etable if isinstance(etable, torch.Tensor) else dev(etable)
        self.nspec, self.ntemp, self.nlayers, self.nwave = self.etable.shape
        self.ttable = dev(ttable)
        self.tmin, self.tmax = float(np.min(ttable)), float(np.max(ttable))
        self.wn = dev(wn)
        self.rt_path, self.itop, self.maxdepth = rt_path, itop, maxdepth
        self.rstar = float(rstar)
        self.set_radius(radius)
        if rt_path != 'transit':
            if quadrature_mu is None:
                quadrature_mu, quadrature_weights = default_quadrature()
            elif quadrature_weights is None:
                raise ValueError('quadrature_mu needs quadrature_weights')
            self.mu = dev(quadrature_mu)
            self.weights = dev(quadrature_weights)
        self.ec = torch.zeros((self.nlayers, self.nwave), dtype=torch.float64, device='cuda')
        if self._pending_order is not None:
            self.set_column_order(self._pending_order)

    def set_column_order(self, order):
        """Work the columns of eval_bands' batches in the order `order` (a permutation of
        range(nwave); None: back to grid order).  A second copy of the table is kept with its
        wavenumber axis in that order, so that every stage still streams contiguous columns."""
        self.tile_limit = None
        if order is None:
            self.column_order = self.etable_ordered = None
            return
        order = torch.as_tensor(order, device='cuda').to(torch.int64).contiguous()
        if order.shape != (self.nwave,) or \
                not bool(torch.equal(torch.sort(order).values,
                                     torch.arange(self.nwave, device='cuda'))):
            raise ValueError('column order: not a permutation of range(nwave)')
        out = torch.empty_like(self.etable)
        for s in range(self.nspec):               # (species by species: a bounded temporary)
            torch.index_select(self.etable[s], -1, order, out=out[s])
        self.etable_ordered = out
        self.column_order = order.to(torch.int32)
        self.wn_ordered = self.wn[order].contiguous()

    def order_columns(self, temp, dens, radius=None):
        """Order the columns by the layer at which the model (temp[L], dens[L, nspec], radius[L])
        becomes optically thick (its ideep, _trapezoid.c:259-273).  The reference stops a column
        there; the matrix-core transit kernel can stop only when all 32 columns of a wavefront
        have -- which neighbours on the wavenumber grid never do together (a line core next to a
        window), and columns of similar depth do: at C5's shape 62 % of the products and 76 % of
        the layer reads remain.  Any model near the ones to come will do (walkers of a retrieval
        differ by per cent); the spectra do not depend on the order, only the time does."""
        temp = (temp if isinstance(temp, torch.Tensor) else dev(temp)).reshape(1, -1)
        dens = (dens if isinstance(dens, torch.Tensor) else dev(dens)).reshape(1, self.nlayers, -1)
        rad = self.radius if radius is None else \
            (radius if isinstance(radius, torch.Tensor) else dev(radius))
        rad = rad.reshape(1, -1).contiguous()
        ec = interp_ec_batch(self.etable, self.ttable, temp.contiguous(), dens.contiguous())
        if self.rt_path == 'transit':
            _, _, ideep = transit_spectrum_batch(ec, transit_path_device(rad, self.itop), rad,
                                                 self.rstar, self.itop, self.nlayers,
                                                 self.maxdepth, want_depth=True)
            ideep = ideep[0]
        else:
            # (emission: a wavefront of the fused kernel walks the layers until its last lane has
            # reached maxdepth -- lanes that stop together waste nothing)
            _, ideep = plane_parallel_optical_depth(
                ec[0], (rad[0, :-1] - rad[0, 1:]).contiguous(), self.itop, self.nlayers,
                self.maxdepth)
        order = torch.sort(ideep, stable=True).indices
        self.set_column_order(order)
        if self.tile_margin >= 0 and self._ordered_supported():
            # The layers nobody reads: walkers of a retrieval cross maxdepth within a layer or two
            # of the base model (measured at C5's shape: -1 ... +2 layers), so a block of 256
            # ordered columns needs the row tiles (transit) / layers (emission) up to the one
            # holding its deepest base crossing + tile_margin layers -- the interpolation writes
            # only those (80 % of ec at C5's shape), and a walker that does run past them is
            # flagged on the device and repaired (see _eval_chunk): the spectra never depend on
            # the limits.
            sorted_ideep = ideep[order].to(torch.int64)
            nblk = -(-self.nwave // 256)
            pad = nblk * 256 - self.nwave
            if pad:
                sorted_ideep = torch.cat([sorted_ideep, sorted_ideep[-1:].expand(pad)])
            bmax = sorted_ideep.view(nblk, 256).max(dim=1).values
            ntiles = -(-(self.nlayers - self.itop) // 16)
            tile = torch.clamp((bmax - self.itop + self.tile_margin) // 16, 0, ntiles - 1)
            # (worth its two gated repair launches only where it saves something: C5's emission
            # geometry crosses maxdepth near the bottom and keeps 97 % of the layers)
            written = torch.clamp(16 * (tile + 1), max=self.nlayers - self.itop).double().mean() / \
                (self.nlayers - self.itop)
            if float(written) <= 0.95 or self.tile_margin == 0:
                self.tile_limit = tile.to(torch.int32).contiguous()

    def set_radius(self, radius):
        self.radius = dev(radius)
        if self.rt_path == 'transit':
            self.raypath = dev(pack_raypath(transit_path(radius, self.itop), self.itop))
        else:
            self.intervals = dev(-np.diff(np.asarray(radius, float)))

    def eval(self, temp, dens, continuum_density=None):
        """temp[L] (K, inside the table's range -- the caller rejects the rest like
        line_sampling.py:426-427), dens[L, nspec] (molecules cm-3) -> spectrum[W].
        With a Continuum attached, continuum_density = {species: n[L]} feeds its terms
        (pyrat/opacity.py:206-257: every model adds to the same ec)."""
        temp_host = None if isinstance(temp, torch.Tensor) else np.asarray(temp, float)
        if temp_host is not None and (np.any(temp_host < self.tmin) or
                                      np.any(temp_host > self.tmax)):
            raise ValueError(f'temperature outside the {self.tmin:.1f}-{self.tmax:.1f} '
                             'K range of the table (the reference rejects such a model, '
                             'line_sampling.py:426-427)')
        self.temp = temp if isinstance(temp, torch.Tensor) else dev(temp)
        dens = dens if isinstance(dens, torch.Tensor) else dev(dens)
        t = self._timer
        if t is not None:
            t.start('extinction')
        interp_ec(self.ec, self.etable, self.ttable, self.temp, dens, 0, self.nlayers,
                  assign=True)
        if self.continuum is not None:
            # the continuum's per-layer factors are prepared on the host: hand it host
            # temperatures when the caller has them (no device -> host copy in the loop)
            self.continuum.add(self.ec, temp_host if temp_host is not None
                               else self.temp.cpu().numpy(), continuum_density)
        if t is not None:
            t.mark('extinction', 'odepth')
        if self.rt_path == 'transit':
            self.spectrum, self.depth, self.ideep = transit_spectrum(
                self.ec, self.raypath, self.radius, self.rstar, self.itop, self.nlayers,
                self.maxdepth)
        else:
            self.depth, self.ideep = plane_parallel_optical_depth(
                self.ec, self.intervals, self.itop, self.nlayers, self.maxdepth)
            if t is not None:
                t.mark('odepth', 'spectrum')
            self.spectrum = emission_flux(self.depth, self.ideep, self.wn, self.temp, self.mu,
                                          self.weights, self.itop)
        if t is not None:
            t.mark('spectrum')
        return self.spectrum

    @property
    def timestamps(self):
        """Seconds of the last eval() by stage: 'extinction' (interpolation of the table +
        continuum terms), 'odepth', 'spectrum' -- the reference's keys (pyrat_obj.py:203-214)."""
        if self._timer is None:
            raise _capi.PbError('this model was built with timestamps=False')
        return self._timer.read()

    def eval_bands(self, temps, dens, bands, radius=None, chunk=64, streams=None,
                   f_dilution=None):
        """Batched-walker evaluation (the inner loop of a retrieval, pyrat_obj.py:225-385
        without the parameter mapping): temps[nw, L], dens[nw, L, nspec] device tensors,
        optional per-walker radius[nw, L] (the hydrostatic profile changes with every model),
        bands: PassBands on this model's grid -> bandflux[nw, nbands].  Every stage is ONE
        launch per chunk of walkers -- interp_ec, transit_path, optical depth + transmission,
        band integration -- with no per-walker Python and no host synchronisation (except the
        one-time column ordering of the first call with column_order='auto': class docstring).  Walkers
        whose temperatures leave the table's range get +inf, like eval()'s reject path
        (pyrat_obj.py:302-320, 378-380).  Emission geometry: f_dilution[nw] = the walkers'
        dilution factors (pyrat_obj.py:296-297), and bands.set_eclipse(...) for the planet-to-star
        flux ratios of an eclipse retrieval (pyrat_obj.py:662-665)."""
        assert self.rt_path in ('transit', 'emission') and self.continuum is None, \
            'eval_bands: transit or emission geometry on sampled cross sections'
        assert f_dilution is None or self.rt_path == 'emission', 'f_dilution: emission geometry'
        assert f_dilution is None or f_dilution.shape == (temps.shape[0],)
        nw = temps.shape[0]
        out = torch.empty((nw, bands.nbands), dtype=torch.float64, device='cuda')
        if radius is None:
            radius = self.radius.view(1, -1)
        shared_radius = radius.shape[0] == 1
        transit = self.rt_path == 'transit'
        if self._auto_order and self.column_order is None and nw > 0 and self.nwave >= 64 and \
                not (transit and self._one_pass()):
            # ONE-TIME set-up of the first batch (class docstring): a host read-back, a sort and a
            # permuted second copy of the table.  Skipped -- grid order, nothing else changes --
            # where the ordered kernels do not exist for the shape, while the stream is being
            # captured into a graph, and when the second copy + this batch's ec would not fit.
            t0 = temps[0]
            table_bytes = self.etable.numel() * 8
            ec_bytes = 8 * min(chunk, nw) * self.nlayers * self.nwave
            if not self._ordered_supported() or torch.cuda.is_current_stream_capturing():
                self._auto_order = self._auto_order and self._ordered_supported()
            elif torch.cuda.mem_get_info()[0] < 1.25 * table_bytes + 2 * ec_bytes:
                self._auto_order = False      # no room for the second copy of the table: grid order
            # (a walker outside the table's range would order by garbage: wait for a valid one)
            elif bool(((t0 >= self.tmin) & (t0 <= self.tmax)).all()):
                self.order_columns(t0, dens[0], radius[0])
        path1 = (transit_path_device(radius[0], self.itop).view(1, -1)
                 if shared_radius and transit else None)
        # Consecutive chunks are independent: with `streams` > 1 (PB_EVAL_STREAMS) chunk i runs on
        # side stream i % streams.  Measured at C5's shape and NOT the default: the interpolation
        # of one chunk beside the optical-depth pass of the previous one gains nothing (two chunks
        # of 32 on two streams 2.97 ms, one chunk of 64 2.73 ms per 64 walkers): both stages
        # stream every walker's ec through HBM.
        nchunks = -(-nw // chunk)
        if streams is None:
            streams = int(os.environ.get('PB_EVAL_STREAMS', '1'))
        streams = max(1, min(streams, nchunks))
        caller = torch.cuda.current_stream()
        if streams > 1:
            self._eval_streams = side_streams(streams)
            for st in self._eval_streams[:streams]:
                st.wait_stream(caller)
        for ci, w0 in enumerate(range(0, nw, chunk)):
            if streams > 1:
                with torch.cuda.stream(self._eval_streams[ci % streams]):
                    self._eval_chunk(temps, dens, bands, radius, shared_radius, path1, out, w0,
                                     min(w0 + chunk, nw), f_dilution)
            else:
                self._eval_chunk(temps, dens, bands, radius, shared_radius, path1, out, w0,
                                 min(w0 + chunk, nw), f_dilution)
        if streams > 1:
            for st in self._eval_streams[:streams]:
                caller.wait_stream(st)
        call('pb_reject_walkers', _ptr(out), _ptr(temps.contiguous()), self.tmin, self.tmax,
             self.nlayers, bands.nbands, nw, _stream())
        return out

    def _ordered_supported(self):
        """Whether the depth-ordered kernels exist for this model's shape: the transit form is the
        matrix-core kernel only (pb_transit_spectrum_ordered: 2 ... 128 impact parameters, i.e.
        2 <= nlayers - itop <= 128); the emission form has no limit."""
        if self.rt_path != 'transit':
            return True
        return 2 <= self.nlayers - self.itop <= 128 and self.nwave >= 2

    def _one_pass(self):
        """The transit batch through pb_table_transit_batch (interpolation, optical depth and
        transmission in one pass, ec never stored): opt-in (`one_pass = True` or
        PB_TABLE_TRANSIT=1).  It saves the ec[walkers, L, W] buffer (4.1 GB per 64 walkers at
        C5's shape); at that shape it runs 2.54 ms per 64 walkers when the walkers resemble one
        another (two walkers per wavefront share their table loads; the two passes: 2.42-2.70 box
        to box), 2.86 when they do not -- which only the device knows, hence not the default."""
        want = getattr(self, 'one_pass', None)
        if want is None:
            want = os.environ.get('PB_TABLE_TRANSIT', '0') == '1'
        return bool(want) and table_transit_supported(self.nspec, self.ntemp, self.nlayers,
                                                      self.itop, self.nlayers, self.nwave)

    def _eval_chunk(self, temps, dens, bands, radius, shared_radius, path1, out, w0, w1,
                    f_dilution=None):
        """One chunk of eval_bands: walkers [w0, w1) through every stage, one launch each."""
        n = w1 - w0
        if self.rt_path == 'transit' and self._one_pass():
            # interpolation + optical depth + transmission in one pass: ec is never stored
            if shared_radius:
                rad = radius.expand(n, -1).contiguous()
                path = path1.expand(n, -1).contiguous()
            else:
                rad = radius[w0:w1].contiguous()
                path = transit_path_device(rad, self.itop)
            spectra = table_transit_batch(self.etable, self.ttable, temps[w0:w1], dens[w0:w1],
                                          path, rad, self.rstar, self.itop, self.nlayers,
                                          self.maxdepth)
            bands.integrate_batch(spectra, out[w0:w1])
            return
        # (an explicit order on a shape the ordered transit kernel does not take -- more than 128
        # impact parameters -- is worked in grid order: the spectra do not depend on the order)
        ordered = self.column_order is not None and self._ordered_supported()
        table = self.etable_ordered if ordered else self.etable
        limited = ordered and self.tile_limit is not None
        if limited:
            # (ec keeps whatever an earlier batch left in the layers that are not written: they are
            # read by no one, or the walker is flagged and repaired)
            flags = torch.zeros(n + 1, dtype=torch.int32, device=table.device)
            iwork = torch.empty(n * self.nlayers * 17 + 8, dtype=torch.float64,
                                device=table.device)
            twork = None
            if self.rt_path == 'transit':
                twork = torch.empty(_capi.lib().pb_transit_work_doubles(
                    self.nlayers, int(self.itop), int(self.nlayers), self.nwave, n),
                    dtype=torch.float64, device=table.device)
            ec = interp_ec_batch(table, self.ttable, temps[w0:w1], dens[w0:w1],
                                 tile_limit=self.tile_limit, row0=self.itop, work=iwork)
        else:
            ec = interp_ec_batch(table, self.ttable, temps[w0:w1], dens[w0:w1])
        if self.rt_path != 'transit':
            rad = radius.expand(n, -1) if shared_radius else radius[w0:w1]
            intervals = (rad[:, :-1] - rad[:, 1:]).contiguous()            # -diff(radius)
            if limited:
                spectra = emission_flux_batch(ec, intervals, self.wn_ordered, temps[w0:w1],
                                              self.mu, self.weights, self.itop, self.nlayers,
                                              self.maxdepth, self.column_order,
                                              tile_limit=self.tile_limit, flags=flags)
                # (device-gated repair, as in the transit branch below)
                interp_ec_batch(table, self.ttable, temps[w0:w1], dens[w0:w1], out=ec,
                                gate=flags[n:n + 1], work=iwork)
                emission_flux_batch(ec, intervals, self.wn_ordered, temps[w0:w1], self.mu,
                                    self.weights, self.itop, self.nlayers, self.maxdepth,
                                    self.column_order, gate=flags, out=spectra)
            else:
                spectra = emission_flux_batch(ec, intervals,
                                              self.wn_ordered if ordered else self.wn,
                                              temps[w0:w1], self.mu, self.weights, self.itop,
                                              self.nlayers, self.maxdepth,
                                              self.column_order if ordered else None)
            bands.integrate_batch(spectra, out[w0:w1],
                                  None if f_dilution is None else f_dilution[w0:w1].contiguous())
            return
        if shared_radius:
            rad = radius.expand(n, -1).contiguous()
            path = path1.expand(n, -1).contiguous()
        else:
            rad = radius[w0:w1].contiguous()
            path = transit_path_device(rad, self.itop)
        if ordered and limited:
            spectra = transit_spectrum_ordered(ec, path, rad, self.column_order, self.rstar,
                                               self.itop, self.nlayers, self.maxdepth,
                                               tile_limit=self.tile_limit, flags=flags, work=twork)
            # repair, gated on the device: the full interpolation if ANY walker ran past its
            # limit, then the transit of the flagged walkers -- two launches of workgroups that
            # return at once otherwise, no host round trip
            interp_ec_batch(table, self.ttable, temps[w0:w1], dens[w0:w1], out=ec,
                            gate=flags[n:n + 1], work=iwork)
            transit_spectrum_ordered(ec, path, rad, self.column_order, self.rstar, self.itop,
                                     self.nlayers, self.maxdepth, gate=flags, out=spectra,
                                     work=twork)
        elif ordered:
            # columns in depth order: wavefronts stop at the row tile where theirs have all crossed
            spectra = transit_spectrum_ordered(ec, path, rad, self.column_order, self.rstar,
                                               self.itop, self.nlayers, self.maxdepth)
        else:
            spectra = transit_spectrum_batch(ec, path, rad, self.rstar, self.itop, self.nlayers,
                                             self.maxdepth)
        bands.integrate_batch(spectra, out[w0:w1])
