"""Drop-in counterparts of the reference's native extension modules `pyratbay.lib.*`
(src_c/*.c, built by the reference's setup.py:22-31): same module names, function names,
positional signatures, in-place mutation and return conventions,
computed on the MI355X through libpbhip.so.  NumPy arrays in, NumPy arrays out; every
call is a host -> device -> host round trip, so these exist for API parity and for
callers that cannot be changed -- the device-resident path is pyratbay_amd.engine.

    import sys, pyratbay_amd.lib as hip
    for name in hip.MODULES:                      # before `import pyratbay`
        sys.modules[f'pyratbay.lib.{name}'] = getattr(hip, name)
"""
from . import (_extcoeff, vprofile, _trapezoid, _simpson, _blackbody, cutils, _indices,
               _alkali, _spline)

MODULES = ['_extcoeff', 'vprofile', '_trapezoid', '_simpson', '_blackbody', 'cutils',
           '_indices', '_alkali', '_spline']
