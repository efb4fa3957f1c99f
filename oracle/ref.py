"""TEST INFRASTRUCTURE ONLY.

Loader for the UNMODIFIED reference C extensions compiled by oracle/Makefile into
oracle/_ref/ (git-ignored binaries; they travel to the GPU box, the reference
sources do not).  `available()` is False when they were never built.
"""
import importlib.util
import glob
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}
NAMES = ['_extcoeff', 'vprofile', '_trapezoid', '_simpson', '_blackbody', 'cutils',
         '_indices']


def _path(name):
    hits = glob.glob(os.path.join(_HERE, '_ref', name + '.*.so'))
    return hits[0] if hits else None


def available():
    return all(_path(n) is not None for n in NAMES)


def module(name):
    if name not in _CACHE:
        path = _path(name)
        if path is None:
            raise ImportError(f'oracle/_ref/{name} not built (run make -C oracle ref)')
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _CACHE[name] = mod
    return _CACHE[name]
