"""pyratbay.lib._simpson (src_c/_simpson.c, include/simpson.h) on the GPU."""
import numpy as np
import torch

from . import _np
from ._np import call, ptr, stream


def geth(h):
    """geth(h) -> [hsum, hratio, hfactor] (src_c/_simpson.c:36-69); O(n) host arithmetic
    on the layer spacing, pairs start at index len(h) % 2."""
    h = _np.f64(h)
    n = len(h)
    if n == 0:
        return [0, 0, 0]
    j = 2 * np.arange(n // 2) + n % 2
    hsum = h[j] + h[j + 1]
    hratio = h[j] / h[j + 1]
    hfactor = hsum * hsum / (h[j] * h[j + 1])
    return [hsum, hratio, hfactor]


def simps2D(y, h, nint, hsum, hratio, hfactor):
    """simps2D(y, h, nint, hsum, hratio, hfactor) -> new [nwave] (src_c/_simpson.c:167-203)"""
    yd = _np.dev(_np.f64(y))
    ny, nwave = yd.shape
    bufs = [_np.dev(_np.f64(a)) for a in (h, hsum, hratio, hfactor)]
    nd = _np.idev(nint)
    out = torch.empty(nwave, dtype=torch.float64, device='cuda')
    call('pb_simps2D', ptr(out), ptr(yd), ny, nwave, ptr(bufs[0]), ptr(nd), ptr(bufs[1]),
         ptr(bufs[2]), ptr(bufs[3]), stream())
    return _np.host(out)


def simps(y, h, hsum, hratio, hfactor):
    """simps(y, h, hsum, hratio, hfactor) -> float (src_c/_simpson.c:104-131), as the
    one-column case of simps2D."""
    y = _np.f64(y)
    if len(y) < 2:
        return 0.0
    return float(simps2D(y.reshape(-1, 1), h, np.array([len(y)], np.int32), hsum, hratio,
                         hfactor)[0])
