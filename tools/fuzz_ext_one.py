import sys, os, traceback
sys.path.insert(0, '.')
import numpy as np
from tools import fuzz_ext
from oracle import ref
from pyratbay_amd import engine, _capi
if os.environ.get('PB_PROBE_LIB'):
    _capi.LIBPATH = os.path.abspath(os.environ['PB_PROBE_LIB'])
for seed in [int(x) for x in sys.argv[1:]]:
    try:
        info = fuzz_ext.one(engine, ref, np.random.default_rng(7000 + seed), seed)
        print('seed', seed, 'ok', info)
    except Exception:
        print('seed', seed, 'FAIL')
        traceback.print_exc(limit=2)
