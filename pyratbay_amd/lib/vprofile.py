"""pyratbay.lib.vprofile (src_c/vprofile.c:42-114) on the GPU."""
import numpy as np

from .. import engine
from . import _np


def grid(profile, psize, index, lorentz, doppler, dwn, verb):
    """grid(profile, psize, index, lorentz, doppler, dwn, verb) -> 1

    Fills `profile` with the concatenated Voigt profiles, resets the zero entries of
    `psize` to the size of the previous Doppler column and writes the start `index` of
    each profile -- all in place, as the reference."""
    size_in = _np.read_int(psize)
    table = engine.VoigtTable.build(_np.f64(lorentz), _np.f64(doppler), size_in, float(dwn),
                                    1, keep_flat=True)
    if profile.dtype != np.float64:
        raise TypeError('profile must be a float64 array')
    if profile.size < table.nprofile:
        raise ValueError(f'profile has {profile.size} samples, the grid needs '
                         f'{table.nprofile}')
    flat = table.flat()
    profile.reshape(-1)[:table.nprofile] = flat
    _np.write_int(psize, table.size)
    _np.write_int(index, table.index)
    table.close()
    return 1
