"""pyratbay.lib._blackbody (src_c/_blackbody.c) on the GPU."""
import numpy as np
import torch

from . import _np
from ._np import call, ptr, stream


def blackbody_wn_2D(wn, temp, B=None, last=None):
    """blackbody_wn_2D(wn, temp[, B[, last]]) -> new B[nlayers, nwave], or 1 when B is
    given and filled in place up to last[i] (src_c/_blackbody.c:35-75)."""
    w = _np.dev(_np.f64(wn))
    t = _np.dev(_np.f64(temp))
    fresh = B is None
    Bd = (torch.empty((t.shape[0], w.shape[0]), dtype=torch.float64, device='cuda')
          if fresh else _np.dev(_np.f64(B)))
    ld = None if last is None else _np.idev(last)
    call('pb_blackbody_wn_2D', ptr(Bd), ptr(w), w.shape[0], ptr(t), t.shape[0], ptr(ld),
         stream())
    if fresh:
        return _np.host(Bd)
    B[...] = _np.host(Bd)
    return 1


def blackbody_wn(wn, temp, B=None):
    """blackbody_wn(wn, temp[, B]) -> new B[nwave] or 1 (src_c/_blackbody.c:98-130)"""
    w = _np.dev(_np.f64(wn))
    Bd = torch.empty(w.shape[0], dtype=torch.float64, device='cuda')
    call('pb_blackbody_wn', ptr(Bd), ptr(w), w.shape[0], float(temp), stream())
    if B is None:
        return _np.host(Bd)
    B[...] = _np.host(Bd)
    return 1
