"""pyratbay.lib._trapezoid (src_c/_trapezoid.c) on the GPU."""
import numpy as np
import torch

from .. import engine
from . import _np
from ._np import call, ptr, stream


def trapezoid(data, intervals):
    """trapezoid(data, intervals) -> float (src_c/_trapezoid.c:28-48).  1-D and tiny:
    evaluated as a one-column trapezoid2D."""
    h = _np.f64(intervals)
    if len(h) < 1:
        return 0.0
    d = _np.f64(data).reshape(-1, 1)
    return float(trapezoid2D(d, h, np.array([len(h)], np.int32))[0])


def trapezoid2D(data, intervals, nint):
    """trapezoid2D(data, intervals, nint) -> new [nwave] (src_c/_trapezoid.c:70-90)"""
    d = _np.dev(_np.f64(data))
    h = _np.dev(_np.f64(intervals))
    n = _np.idev(nint)
    nrows, nwave = d.shape
    out = torch.empty(nwave, dtype=torch.float64, device='cuda')
    call('pb_trapezoid2D', ptr(out), ptr(d), ptr(h), ptr(n), nrows, nwave, stream())
    return _np.host(out)


def cumulative_sum(output, data, intervals, threshold):
    """cumulative_sum(output, data, intervals, threshold) -> int
    (src_c/_trapezoid.c:114-147).  A 1-D running sum with early exit: host-side scalar
    work (no in-package caller in the reference)."""
    h = _np.f64(intervals)
    d = _np.f64(data)
    output[0] = 0.0
    if len(h) < 1:
        return 0
    for i in range(len(h)):
        output[i + 1] = output[i] + 0.5 * h[i] * (d[i + 1] + d[i])
        if output[i + 1] >= threshold:
            return i + 1
    return len(h)


def plane_parallel_optical_depth(depth, ideep, extinction, intervals, maxdepth, itop,
                                 ibottom):
    """plane_parallel_optical_depth(depth, ideep, extinction, intervals, maxdepth, itop,
    ibottom) -> None (src_c/_trapezoid.c:175-213); depth and ideep written in place."""
    ec = _np.dev(_np.f64(extinction))
    d = _np.dev(_np.f64(depth))
    h = _np.dev(_np.f64(intervals))
    nlayers, nwave = d.shape
    idp = torch.empty(nwave, dtype=torch.int32, device='cuda')
    call('pb_plane_parallel_optical_depth', ptr(d), ptr(idp), ptr(ec), ptr(h),
         float(maxdepth), int(itop), int(ibottom), nlayers, nwave, stream())
    depth[...] = _np.host(d)
    _np.write_int(ideep, _np.host(idp))
    return None


def optdepth(data, intervals, taumax, ideep, ilay):
    """optdepth(data, intervals, taumax, ideep, ilay) -> new tau[nwave]
    (src_c/_trapezoid.c:238-276); ideep updated in place."""
    d = _np.dev(_np.f64(data))
    h = _np.dev(_np.f64(intervals))
    idp = _np.idev(ideep)
    nwave = d.shape[1]
    tau = torch.empty(nwave, dtype=torch.float64, device='cuda')
    call('pb_optdepth', ptr(tau), ptr(d), nwave, ptr(h), h.shape[0], float(taumax), ptr(idp),
         int(ilay), nwave, stream())
    _np.write_int(ideep, _np.host(idp))
    return _np.host(tau)


def intensity(tau, ideep, planck, mu, rtop):
    """intensity(tau, ideep, planck, mu, rtop) -> new [nmu, nwave]
    (src_c/_trapezoid.c:304-341)"""
    out = engine.intensity(_np.dev(_np.f64(tau)), _np.idev(ideep), _np.dev(_np.f64(planck)),
                           _np.dev(_np.f64(mu)), int(rtop))
    return _np.host(out)
