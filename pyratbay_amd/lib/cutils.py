"""pyratbay.lib.cutils (src_c/cutils.c) on the GPU / host."""
import numpy as np
import torch

from . import _np
from ._np import call, ptr, stream


def ediff(arr):
    """ediff(arr) -> new [n-1] consecutive differences (src_c/cutils.c:27-42)"""
    a = _np.dev(_np.f64(arr))
    n = a.shape[0]
    out = torch.empty(max(n - 1, 0), dtype=torch.float64, device='cuda')
    call('pb_ediff', ptr(out), ptr(a), n, stream())
    return _np.host(out)


def arrbinsearch(values, array):
    """arrbinsearch(values, array) -> int32[n] index of the closest element
    (src_c/cutils.c:63-80).  No in-package caller in the reference; host-side bisection
    with the reference's tie rule (utils.h:75-89), searching [0, n-1] (the reference
    passes hi=n, one past the end)."""
    values = _np.f64(values)
    array = _np.f64(array)
    out = np.empty(len(values), np.int32)
    for k, v in enumerate(values):
        lo, hi = 0, len(array) - 1
        while hi - lo > 1:
            mid = (hi + lo) // 2
            if array[mid] > v:
                hi = mid
            else:
                lo = mid
        out[k] = hi if abs(array[hi] - v) < abs(array[lo] - v) else lo
    return out
