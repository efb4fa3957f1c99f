"""NumPy <-> device plumbing shared by the drop-in modules."""
import ctypes

import numpy as np
import torch

from .. import engine
from .._capi import call                      # noqa: F401  (re-export)

dev = engine.dev
ptr = engine._ptr
stream = engine._stream


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def int32_view(a):
    """The reference reads and writes C `int` (32 bit) at the array's byte strides whatever
    the NumPy dtype is (src_c/include/ind.h:31-37): for int64 arrays that is the low word
    of each element (little endian).  Returns an int32 array aliasing those words."""
    a = np.asarray(a)
    if a.dtype.kind not in 'iu' or a.dtype.itemsize not in (4, 8):
        raise TypeError(f'integer array expected, got {a.dtype}')
    if a.dtype.itemsize == 4:
        return a.view(np.int32)
    if a.size == 0:
        return np.zeros(a.shape, np.int32)
    if any(s < 0 for s in a.strides):
        raise ValueError('negative strides are not supported')
    extent = sum((n - 1) * s for n, s in zip(a.shape, a.strides)) + a.dtype.itemsize
    buf = (ctypes.c_char * extent).from_address(a.ctypes.data)
    # the view aliases a's memory: callers use it immediately, while `a` is alive
    return np.ndarray(a.shape, np.int32, buffer=buf, strides=a.strides)


def read_int(a):
    """Contiguous int32 copy of what the reference would read from `a`."""
    return np.ascontiguousarray(int32_view(a))


def write_int(dest, values):
    """Store `values` the way the reference does: 32-bit writes at dest's strides."""
    int32_view(dest)[...] = values


def host(t):
    return t.cpu().numpy()


def idev(a):
    return dev(read_int(a), torch.int32)
