#!/usr/bin/env python3
"""Fixture G17: what the reference's OWN callers hand to its native modules (VERDICT round 4,
item 5).  A recording proxy stands in for every module of `pyratbay.lib` while the real package
runs `pb.run()` (transit, emission) on the mock H2O line list; every call made by the package's
Python layer -- pyrat/voigt.py, pyrat/extinction.py:170-213, opacity/optic_depth.py:89-136,
spectrum/radiative_transfer.py, spectrum/spec_tools.py ... -- is stored with

    module, function, per argument: kind (array | scalar | None), and for arrays the dtype,
    shape, BYTE STRIDES, C-contiguity, writeable flag and the values;
    the return value; the post-call values of every array argument the call changed in place.

Arrays are stored once per distinct content (the Voigt table, the fine grid, the line list repeat
in every one of the 51 per-layer calls).  Build container only (needs /root/reference):

    python tests/golden/make_golden_trace.py

tests/test_gpu_dropin_trace.py replays the trace through `pyratbay_amd.lib.*` with arrays rebuilt
at the recorded dtypes and strides.  Only data is written (numbers and dtype/stride metadata).
"""
import hashlib
import json
import os
import pickle
import shutil
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_e2e import reference_package, REF, CFG_TLI      # noqa: E402

LIB_MODULES = ('_extcoeff', 'vprofile', '_trapezoid', '_simpson', '_blackbody', 'cutils',
               '_indices', '_alkali', '_spline', '_pt')

# a small run: 1.000-1.004 um (~40 cm-1, 400 samples), narrow Voigt table -- every call of the
# path is made, the arrays stay small
CFG_SPEC = '''
[pyrat]
runmode = spectrum
logfile = {work}/trace_{rt}.log
rt_path = {rt}
atmfile = {ref}/tests/inputs/atmosphere_uniform_test.atm
tlifile = {work}/mock_h2o.tli
radmodel = hydro_m
wl_low = 1.000 um
wl_high = 1.004 um
wnstep = 0.1
wnosamp = 24
voigt_extent = 25.0
voigt_cutoff = 4.0
nlor = 10
ndop = 6
rstar = 1.27 rsun
tstar = 5800.0
mplanet = 0.6 mjup
rplanet = 1.0 rjup
refpressure = 0.1 bar
maxdepth = 10.0
ncpu = 1
verb = 0
{extra}
'''


class Recorder:
    """Calls are written to `spool` one file each (the extinction loop runs in forked children,
    pyrat/line_by_line.py:232-246: their records would die with them) and merged by collect()."""

    def __init__(self, spool):
        self.spool = spool
        os.makedirs(spool, exist_ok=True)
        self.calls = []
        self.arrays = {}            # digest -> array (stored once)

    @staticmethod
    def key(a):
        a = np.ascontiguousarray(a)
        return hashlib.sha1(a.tobytes() + str(a.dtype).encode() +
                            str(a.shape).encode()).hexdigest()[:16]

    def describe(self, v, arrays):
        if v is None:
            return {'kind': 'none'}
        if isinstance(v, np.ndarray):
            k = self.key(v)
            arrays[k] = np.ascontiguousarray(v).copy()
            return {'kind': 'array', 'dtype': v.dtype.str, 'shape': list(v.shape),
                    'strides': list(v.strides), 'c_contiguous': bool(v.flags.c_contiguous),
                    'writeable': bool(v.flags.writeable), 'values': k}
        if isinstance(v, (bool, np.bool_)):
            return {'kind': 'bool', 'value': bool(v)}
        if isinstance(v, (int, np.integer)):
            return {'kind': 'int', 'value': int(v)}
        if isinstance(v, (float, np.floating)):
            return {'kind': 'float', 'value': float(v), 'hex': float(v).hex()}
        if isinstance(v, (list, tuple)):
            return {'kind': 'list', 'items': [self.describe(x, arrays) for x in v]}
        raise TypeError(f'unrecorded argument type {type(v)}')

    def wrap(self, module, name, fn, context):
        def recorded(*args):
            arrays = {}
            before = [self.describe(a, arrays) for a in args]
            ret = fn(*args)
            after = {}
            for i, a in enumerate(args):
                if isinstance(a, np.ndarray):
                    k = self.key(a)
                    if k != before[i]['values']:
                        after[str(i)] = k
                        arrays[k] = np.ascontiguousarray(a).copy()
            call = {'run': context['run'], 'module': module, 'function': name, 'args': before,
                    'return': self.describe(ret, arrays), 'after': after}
            with open(os.path.join(self.spool, f'{time.time_ns():020d}_{os.getpid()}.pkl'),
                      'wb') as f:
                pickle.dump((call, arrays), f)
            return ret
        recorded.__name__ = name
        return recorded

    def collect(self, keep):
        """Merge the spooled calls in time order; keep(call, nth call of that function in its
        run) decides which ones stay (thinning of the per-layer loops)."""
        nth = {}
        for fname in sorted(os.listdir(self.spool)):
            with open(os.path.join(self.spool, fname), 'rb') as f:
                call, arrays = pickle.load(f)
            k = (call['run'], call['module'], call['function'])
            nth[k] = nth.get(k, -1) + 1
            call['nth'] = nth[k]
            if keep(call, nth[k]):
                self.calls.append(call)
                self.arrays.update(arrays)
            os.remove(os.path.join(self.spool, fname))


def keep(call, nth):
    """Thinning: the per-layer loops repeat one call shape 51 times.  optdepth's `data` is a
    growing slice of ec (4 MB over the loop): a few layers; extinction: every layer of the transit
    run (its arguments repeat, only density / Z / the output row change), every 6th elsewhere."""
    fn = call['function']
    if fn == 'optdepth':
        return nth in (0, 1, 2, 7, 16, 33, 50)
    if fn == 'extinction':
        return call['run'] == 'transit' or nth % 6 == 0
    return True


def install(pb, rec, context):
    """Replace every reference to a native module held by a loaded pyratbay module (the package
    binds them at import: `from ..lib import _extcoeff as ec`) with a recording proxy."""
    import pyratbay.lib as lib
    proxies = {}
    for modname in LIB_MODULES:
        real = getattr(lib, modname, None)
        if real is None:
            continue
        proxy = types.ModuleType(real.__name__)
        for name in dir(real):
            obj = getattr(real, name)
            if callable(obj) and not name.startswith('_'):
                setattr(proxy, name, rec.wrap(modname, name, obj, context))
            elif not name.startswith('__'):
                setattr(proxy, name, obj)
        proxies[id(real)] = proxy
    n = 0
    for mname, mod in list(sys.modules.items()):
        if not mname.startswith('pyratbay') or mod is None:
            continue
        for attr, val in list(vars(mod).items()):
            if isinstance(val, types.ModuleType) and id(val) in proxies:
                setattr(mod, attr, proxies[id(val)])
                n += 1
            elif callable(val) and getattr(val, '__module__', None) and \
                    getattr(val, '__self__', None) is not None and \
                    id(getattr(val, '__self__')) in proxies:
                # `from ..lib._indices import ifirst`: a builtin bound to its module
                real_mod = getattr(val, '__self__')
                setattr(mod, attr, getattr(proxies[id(real_mod)], val.__name__))
                n += 1
    return n


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_trace_')
    try:
        pb = reference_package(work)

        def run(text, name, **kw):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(text.format(work=work, ref=REF, **kw))
            return pb.run(cfg)

        run(CFG_TLI, 'tli')
        rec = Recorder(os.path.join(work, 'spool'))
        context = {'run': None}
        nrebound = install(pb, rec, context)
        print('rebound', nrebound, 'references to native modules')
        spectra = {}
        for rt, extra in (('transit', ''), ('emission', ''),
                          ('emission', 'quadrature = 3')):
            context['run'] = rt + ('_gauss' if extra else '')
            pyrat = run(CFG_SPEC, 'trace_' + context['run'], rt=rt, extra=extra)
            spectra[context['run']] = pyrat.spec.spectrum.copy()
            rec.collect(keep)
            print(context['run'], 'W', pyrat.spec.nwave, 'calls kept so far', len(rec.calls))
        by_fn = {}
        for c in rec.calls:
            k = f"{c['module']}.{c['function']}"
            by_fn[k] = by_fn.get(k, 0) + 1
        print(json.dumps(by_fn, indent=1))
        meta = {'calls': rec.calls, 'counts': by_fn}
        out = {'arr_' + k: v for k, v in rec.arrays.items()}
        out['trace_json'] = np.array(json.dumps(meta))
        for k, v in spectra.items():
            out['spectrum_' + k] = v
        path = os.path.join(HERE, 'g17_call_trace.npz')
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, 'KiB,', len(rec.arrays), 'distinct arrays')
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
