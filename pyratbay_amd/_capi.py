"""ctypes binding of libpbhip.so (include/pbhip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  torch is used only as plumbing for device memory and streams.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PB_LIBPBHIP=<path>: another build of the library -- libpbhip_exp.so (`make EXPERIMENTS=1`: the
# product + the measured dead ends) or one of the debug builds of tools/debug/
LIBPATH = os.environ.get('PB_LIBPBHIP') or os.path.join(_HERE, 'libpbhip.so')

PB_OK = 0


class PbError(RuntimeError):
    pass


_lib = None

vp = C.c_void_p
i32, i64, f64 = C.c_int, C.c_int64, C.c_double

# name: argtypes (all functions return int unless listed in _RESTYPES)
_PROTOS = {
    'pb_version': [],
    'pb_device_count': [C.POINTER(C.c_int)],
    'pb_set_device': [i32],
    'pb_timer_create': [C.POINTER(vp), i32],
    'pb_timer_start': [vp, C.c_char_p, vp],
    'pb_timer_mark': [vp, C.c_char_p, C.c_char_p, vp],
    'pb_timer_count': [vp, C.POINTER(i32)],
    'pb_timer_read': [vp, i32, C.c_char_p, i32, C.POINTER(f64)],
    'pb_timer_destroy': [vp],
    'pb_range_push': [C.c_char_p],
    'pb_range_pop': [],
    'pb_roctx_available': [],
    'pb_voigt_create': [C.POINTER(vp), vp, i32, vp, i32, vp, f64, i32, i32, vp],
    'pb_voigt_from_flat': [C.POINTER(vp), vp, i64, vp, i32, vp, i32, vp, vp, i32, i32, vp],
    'pb_voigt_meta': [vp, vp, vp, C.POINTER(i64)],
    'pb_voigt_flat_to_host': [vp, vp, i64],
    'pb_voigt_device_bytes': [vp],
    'pb_voigt_destroy': [vp],
    'pb_lines_create': [C.POINTER(vp), vp, vp, vp, vp, i64, i32, vp, i64, f64, f64],
    'pb_lines_stats': [vp, C.POINTER(i64 * 3)],
    'pb_lines_grouped_on_device': [vp, C.POINTER(i32)],
    'pb_lines_groups': [vp, vp, vp, vp, vp],
    'pb_lines_destroy': [vp],
    'pb_lbl_create': [C.POINTER(vp), vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp,
                      i32, f64, f64, i32, i32],
    'pb_lbl_set_isoiext': [vp, vp],
    'pb_lbl_set_ethresh': [vp, f64],
    'pb_lbl_set_gather_mode': [vp, i32],
    'pb_lbl_set_concurrency': [vp, i32],
    'pb_lbl_set_record_budget': [vp, i64],
    'pb_lbl_last_chunks': [vp, C.POINTER(i32)],
    'pb_lbl_last_gather_mode': [vp, C.POINTER(i32)],
    'pb_lbl_extinction': [vp, vp, i64, i64, vp, vp, vp, i64, i64, i32, i32, vp],
    'pb_lbl_extinction_begin': [vp, vp, i64, i64, vp, vp, vp, i64, i64, i32, i32, vp],
    'pb_lbl_kmax_buffer': [vp, C.POINTER(vp), C.POINTER(i64)],
    'pb_lbl_extinction_end': [vp, vp],
    'pb_lbl_last_state': [vp, vp, vp, i32, i32, vp],
    'pb_lbl_last_layer_kinds': [vp, vp, vp, i32, vp],
    'pb_lbl_last_work': [vp, C.POINTER(i64 * 3), vp],
    'pb_lbl_last_table_samples': [vp, C.POINTER(i64), vp],
    'pb_lbl_timing_begin': [vp, i32],
    'pb_lbl_timing_end': [vp, C.POINTER(f64), C.POINTER(i32)],
    'pb_lbl_destroy': [vp],
    'pb_interp_ec': [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    'pb_interp_ec_set': [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    'pb_transit_path': [vp, vp, i32, i32, i32, vp],
    'pb_iso_partition': [vp, i64, i64, vp, i64, vp, i32, vp, i32, vp, vp],
    'pb_resample_cross_section': [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32,
                                  i32, vp],
    'pb_interp_ec_batch': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'pb_transit_work_doubles': [i32, i32, i32, i32, i32],
    'pb_transit_spectrum_batch': [vp, vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, i32, i32, vp, vp],
    'pb_transit_spectrum_ordered': [vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, i32, i32, vp, vp],
    'pb_interp_ec_batch_limited': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp],
    'pb_transit_spectrum_limited': [vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, i32, i32, vp, vp,
                                    vp, vp, vp],
    'pb_emission_flux_limited': [vp, vp, vp, vp, vp, vp, vp, vp, i32, f64, i32, i32, i32, i32, i32,
                                 vp, vp, vp, vp],
    'pb_emission_flux_batch': [vp, vp, vp, vp, vp, vp, vp, i32, f64, i32, i32, i32, i32, i32, vp],
    'pb_emission_flux_ordered': [vp, vp, vp, vp, vp, vp, vp, vp, i32, f64, i32, i32, i32, i32, i32, vp],
    'pb_band_integrate_batch': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    'pb_emission_observables': [vp, vp, vp, vp, vp, i64, i32, i32, f64, f64, vp],
    'pb_band_scale': [vp, vp, vp, i32, i32, vp],
    'pb_reject_walkers': [vp, vp, f64, f64, i32, i32, i32, vp],
    'pb_optdepth': [vp, vp, i64, vp, i32, f64, vp, i32, i32, vp],
    'pb_optical_depth_transit': [vp, vp, vp, vp, i32, i32, f64, i32, i32, vp],
    'pb_transit_spectrum': [vp, vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, i32, vp],
    'pb_plane_parallel_optical_depth': [vp, vp, vp, vp, f64, i32, i32, i32, i32, vp],
    'pb_trapezoid2D': [vp, vp, vp, vp, i32, i32, vp],
    'pb_transmission': [vp, vp, vp, vp, i32, f64, i32, i32, vp],
    'pb_blackbody_wn_2D': [vp, vp, i32, vp, i32, vp, vp],
    'pb_blackbody_wn': [vp, vp, i32, f64, vp],
    'pb_intensity': [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'pb_emission_flux': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'pb_transmission_deck': [vp, vp, vp, vp, i32, f64, i32, f64, i32, i32, vp],
    'pb_transit_spectrum_deck': [vp, vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, f64, i32, i32, vp],
    'pb_emission_flux_deck': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    'pb_continuum': [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    'pb_alkali_cross_section': [vp, vp, vp, vp, vp, f64, f64, f64, f64, f64, vp, vp, i32, vp, i32, i32, vp],
    'pb_loglike': [vp, vp, vp, vp, i32, i32, vp],
    'pb_two_stream': [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    'pb_internal_flux': [vp, vp, f64, i32, vp],
    'pb_simps2D': [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp],
    'pb_ediff': [vp, vp, i32, vp],
    'pb_band_integrate': [vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, vp],
}
# entries of libpbhip_exp.so only (include/pbhip.h, "Experiments"): bound when the loaded library has them
_EXP_PROTOS = {
    'pb_lbl_last_wave_layers': [vp, vp, i32, vp],
    'pb_lbl_dyn_stats': [vp, vp],
    'pb_lbl_set_dyn_predict': [vp, i32],
    'pb_table_transit_supported': [i32, i32, i32, i32, i32, i32],
    'pb_table_transit_work_doubles': [i32, i32, i32, i32, i32],
    'pb_table_transit_batch': [vp, vp, vp, vp, vp, vp, vp, f64, i32, i32, f64, i32, i32, i32, i32,
                               i32, vp, vp],
}
_RESTYPES = {'pb_transit_work_doubles': C.c_int64, 'pb_table_transit_work_doubles': C.c_int64,
             'pb_table_transit_supported': C.c_int, 'pb_voigt_destroy': None, 'pb_lines_destroy': None, 'pb_lbl_destroy': None,
             'pb_voigt_device_bytes': i64, 'pb_timer_destroy': None}
_NO_CHECK = set(_RESTYPES) | {'pb_version', 'pb_roctx_available'}


def exported_names():
    return sorted(list(_PROTOS) + ['pb_last_error'])


def experiments():
    """True when the loaded library is the experiments build (libpbhip_exp.so)."""
    return hasattr(lib(), 'pb_table_transit_batch')


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  torch bundles its own libamdhip64 (SONAME
    libamdhip64.so.7, file name libamdhip64.so); libpbhip.so needs the same SONAME.  If
    libpbhip were loaded first it would pull /opt/rocm's copy and torch would later add
    its own: two runtimes, and device pointers / streams handed across would break.
    Loading torch's copy first makes the dynamic loader reuse it for libpbhip."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load libpbhip.so or fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise PbError(
                f'{LIBPATH} is missing: build it with `make -C pyratbay_amd/csrc` '
                '(or __graft_entry__.build()); there is no CPU fallback')
        _preload_torch_hip_runtime()
        handle = C.CDLL(LIBPATH)
        handle.pb_last_error.restype = C.c_char_p
        handle.pb_last_error.argtypes = []
        for name, args in _PROTOS.items():
            fn = getattr(handle, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        for name, args in _EXP_PROTOS.items():
            fn = getattr(handle, name, None)
            if fn is not None:
                fn.argtypes = args
                fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = handle
        global _owner_pid
        _owner_pid = os.getpid()
    return _lib


_owner_pid = None


def call(name, *args):
    # The reference parallelises with fork (pyrat/line_by_line.py:232-246, ncpu > 1).  A
    # forked child of a process that has initialised HIP cannot use the GPU (and on this
    # platform must not try): fail loudly instead of hanging or faulting.
    if _owner_pid is not None and os.getpid() != _owner_pid:
        raise PbError(f'{name}: called in a forked child (pid {os.getpid()}) of the process '
                      f'that initialised the GPU (pid {_owner_pid}); the HIP path needs '
                      'ncpu = 1 -- it batches all layers in one call instead of forking')
    fn = getattr(lib(), name, None)
    if fn is None:
        raise PbError(f'{name} is an experiment that libpbhip.so does not carry: build '
                      '`make -C pyratbay_amd/csrc EXPERIMENTS=1` and set '
                      'PB_LIBPBHIP=pyratbay_amd/libpbhip_exp.so')
    rc = fn(*args)
    if name not in _NO_CHECK and rc != PB_OK:
        raise PbError(f'{name} failed ({rc}): {lib().pb_last_error().decode()}')
    return rc


def hptr(a):
    """Host pointer of a C-contiguous NumPy array (or None)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data_as(vp)


def f64h(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32h(a):
    return np.ascontiguousarray(a, dtype=np.int32)
