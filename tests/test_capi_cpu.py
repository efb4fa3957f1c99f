"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol that
include/pbhip.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(experiments=False):
    """Entry points include/pbhip.h declares: the product's, or those of its `#ifdef
    PB_EXPERIMENTS` section (libpbhip_exp.so only)."""
    text = open(os.path.join(ROOT, 'include', 'pbhip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    exp = re.search(r'#ifdef PB_EXPERIMENTS(.*?)#endif', text, flags=re.S)
    assert exp, 'pbhip.h: no experiments section'
    text = exp.group(1) if experiments else text.replace(exp.group(0), '')
    return sorted(set(re.findall(r'\b(pb_[a-zA-Z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    """libpbhip.so exports exactly what the header declares outside its experiments section --
    nothing missing, nothing more (`nm -D`) -- and the Python binding covers all of it."""
    import subprocess
    from pyratbay_amd import _capi
    lib = _capi.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f'{name} declared in pbhip.h but not exported'
    assert set(names) == set(_capi.exported_names())
    so = os.path.join(ROOT, 'pyratbay_amd', 'libpbhip.so')
    if os.path.abspath(_capi.LIBPATH) == so:
        nm = subprocess.run(['nm', '-D', '--defined-only', so], capture_output=True, text=True,
                            check=True).stdout
        exported = sorted(set(re.findall(r' T (pb_[a-zA-Z0-9_]+)$', nm, flags=re.M)))
        assert exported == names, sorted(set(exported) ^ set(names))
        # the measured dead ends are not in the product library
        for name in declared_symbols(experiments=True):
            assert not hasattr(lib, name), f'{name}: an experiment exported by libpbhip.so'
    assert sorted(_capi._EXP_PROTOS) == declared_symbols(experiments=True)


def test_version_and_error_channel():
    from pyratbay_amd import _capi
    assert _capi.call('pb_version') >= 100
    # argument validation happens before any HIP call
    with pytest.raises(_capi.PbError, match='null'):
        _capi.call('pb_lines_stats', None, None)
    assert b'null' in _capi.lib().pb_last_error()


def test_no_gpu_fails_loudly():
    import torch
    from pyratbay_amd import _capi, engine
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(_capi.PbError):
        engine.require_gpu()
    with pytest.raises(_capi.PbError):
        engine.VoigtTable.build([1e-3], [1e-2], [[5]], 1e-3, 4)


def test_product_does_not_import_oracle():
    """The product path must never route through oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, 'pyratbay_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.lower(), os.path.join(dirpath, f)
