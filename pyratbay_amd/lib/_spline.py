"""pyratbay.lib._spline (src_c/_spline.c): the three functions the package calls
(opacity/cia.py:95-101, 151-155).  `lin_interp_2D`, the per-evaluation one, runs on the GPU
through pb_continuum; `second_deriv` and `splinterp_1D` are initialisation-time host code
(pyratbay_amd.continuum)."""
import ctypes as C

import numpy as np

from . import _np
from ._np import call, ptr, stream
from .._capi import hptr
from ..continuum import second_deriv as _second_deriv, splinterp_1D as _splinterp_1D


def second_deriv(yin, xin):
    """second_deriv(yin, xin) -> y2nd (src_c/_spline.c:25-74)."""
    return _second_deriv(_np.f64(yin), _np.f64(xin))


def splinterp_1D(yin, xin, y2nd, xout, extrap):
    """splinterp_1D(yin, xin, y2nd, xout, extrap) -> yout (src_c/_spline.c:95-131)."""
    return _splinterp_1D(_np.f64(yin), _np.f64(xin), _np.f64(y2nd), _np.f64(xout), float(extrap))


def lin_interp_2D(yin, xin, dy_dx, xout, yout, lo, hi):
    """lin_interp_2D(yin, xin, dy_dx, xout, yout, lo, hi): yout[i, lo:hi] = linear
    interpolation of yin[:, lo:hi] along its first axis at xout[i]; returns 0.0, or NaN and
    leaves yout untouched when an xout lies outside xin (src_c/_spline.c:219-260).  dy_dx is
    accepted for the signature; the slopes are formed from yin like the caller's
    (cia.py:112-116)."""
    xin, xout = _np.f64(xin), _np.f64(xout)
    if np.any(xout < xin[0]) or np.any(xout > xin[-1]):
        return float('nan')
    yin = _np.f64(yin)
    nout, nwave = yout.shape
    tab, temps = _np.dev(yin), _np.dev(xin)
    # the kernel accumulates: start from zero, factor 1, then copy the columns it owns
    acc = _np.dev(np.zeros((nout, nwave)))
    ones = _np.dev(np.ones(nout))
    tabs = (C.c_void_p * 1)(tab.data_ptr())
    tmps = (C.c_void_p * 1)(temps.data_ptr())
    one = np.array([len(xin)], np.int32)
    call('pb_continuum', ptr(acc), None, ptr(_np.dev(xout)), nout, nwave, 0, None, None, 1,
         C.cast(tabs, C.c_void_p), C.cast(tmps, C.c_void_p), hptr(one),
         hptr(np.array([lo], np.int32)), hptr(np.array([hi], np.int32)), ptr(ones), None, None,
         None, stream())
    yout[:, lo:hi] = _np.host(acc)[:, lo:hi]
    return 0.0
