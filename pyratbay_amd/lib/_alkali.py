"""pyratbay.lib._alkali (src_c/_alkali.c) on the GPU."""
import numpy as np

from . import _np
from ._np import call, ptr, stream
from .._capi import hptr


def alkali_cross_section(pressure, wn, temp, voigt_det, ec, detuning_wn, mass, lorentz_par,
                         part_func, cutoff, wn0, gf, dwave, i_wn0):
    """alkali_cross_section(pressure, wn, temp, voigt_det, ec, detuning, mass, lpar, Z,
    cutoff, wn0, gf, dwave, i_wn0) -> 1; ec[nlayers,nwave] += cross section (cm2 molec-1)
    (src_c/_alkali.c:30-106).  `dwave` and `i_wn0` are parsed and unused, as there."""
    nlayers, nwave = np.shape(ec)
    ec_d = _np.dev(_np.f64(ec))
    p_d, w_d, t_d = _np.dev(_np.f64(pressure)), _np.dev(_np.f64(wn)), _np.dev(_np.f64(temp))
    vd_d = _np.dev(_np.f64(voigt_det))
    wn0, gf = _np.f64(wn0), _np.f64(gf)
    call('pb_alkali_cross_section', ptr(ec_d), ptr(p_d), ptr(w_d), ptr(t_d), ptr(vd_d),
         float(detuning_wn), float(mass), float(lorentz_par), float(part_func), float(cutoff),
         hptr(wn0), hptr(gf), len(wn0), None, nlayers, nwave, stream())
    ec[...] = _np.host(ec_d)
    return 1
