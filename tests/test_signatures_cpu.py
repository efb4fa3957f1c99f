"""The drop-in modules pyratbay_amd.lib.* against the reference's own call signatures: fixture
G12 holds, for every native module of the reference, the `PyArg_ParseTuple` format of each
function (tests/golden/make_golden_signatures.py reads them from src_c/*.c).  A caller of
`pyratbay.lib.<module>.<function>` passes positional arguments only (METH_VARARGS), so what
must agree is the count of required and optional positional parameters.  CPU only."""
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIG = json.load(open(os.path.join(HERE, 'golden', 'g12_signatures.json')))

# outside SURVEY 8's hot path: temperature-profile models (atmosphere/tmodels) and a 2D spline
# the package never calls (grep: no use under pyratbay/)
NOT_ON_PATH = {('_pt', 'guillot'), ('_pt', 'madhu'), ('_spline', 'splinterp_2D'),
               ('_pt', 'isothermal'), ('_pt', 'tcea')}


def arity(fn):
    ps = list(inspect.signature(fn).parameters.values())
    pos = [p for p in ps if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    req = sum(1 for p in pos if p.default is inspect.Parameter.empty)
    return req, len(pos) - req


def test_every_path_function_has_the_reference_arity():
    import pyratbay_amd.lib as hip
    checked = 0
    for module, funcs in SIG.items():
        for name, info in funcs.items():
            assert info['flags'] == 'METH_VARARGS', (module, name)
            if (module, name) in NOT_ON_PATH or module == '_pt':
                continue
            mod = getattr(hip, module, None)
            assert mod is not None, f'drop-in module {module} missing'
            fn = getattr(mod, name, None)
            assert fn is not None, f'{module}.{name} missing (reference format {info["format"]})'
            assert arity(fn) == (info['required'], info['optional']), \
                f'{module}.{name}: reference "{info["format"]}", drop-in {arity(fn)}'
            checked += 1
    assert checked >= 23
    assert set(hip.MODULES) >= set(SIG) - {'_pt'}


@pytest.mark.parametrize('module,name,fmt', [
    ('_extcoeff', 'extinction', 'OOOOOOOOOOOOOOOOOOOOOdddi|ii'),
    ('_extcoeff', 'interp_ec', 'OOOOOii'),
    ('_trapezoid', 'optdepth', 'OOdOi'),
    ('_trapezoid', 'plane_parallel_optical_depth', 'OOOOdii'),
    ('_trapezoid', 'trapezoid2D', 'OOO'),
    ('_trapezoid', 'intensity', 'OOOOi'),
    ('_blackbody', 'blackbody_wn_2D', 'OO|OO'),
    ('vprofile', 'grid', 'OOOOOdi'),
])
def test_formats_survey_8b_cites(module, name, fmt):
    """The formats SURVEY.md section 8(b) quotes are the ones in the fixture."""
    assert SIG[module][name]['format'] == fmt
