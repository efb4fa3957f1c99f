"""pyratbay.lib._indices (src_c/_indices.c): host-side helpers (SURVEY.md N7 -- one call
per run on an nlayers-long flag array, pyratbay/pyrat/atmosphere.py:517)."""
import numpy as np

from . import _np


def ifirst(data, default_ret=-1):
    """ifirst(data[, default_ret=-1]) -> first index where data == 1
    (src_c/_indices.c:42-59)"""
    hits = np.flatnonzero(_np.read_int(data) == 1)
    return int(hits[0]) if len(hits) else int(default_ret)


def ilast(data, default_ret=-1):
    """ilast(data[, default_ret=-1]) -> last index where data == 1
    (src_c/_indices.c:92-109)"""
    hits = np.flatnonzero(_np.read_int(data) == 1)
    return int(hits[-1]) if len(hits) else int(default_ret)
