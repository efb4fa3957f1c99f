import os, sys
sys.path.insert(0, '.')
import torch, bench
from pyratbay_amd import engine, _capi
from pyratbay_amd.dist import shard_bounds
_capi.LIBPATH = os.path.abspath(os.environ['PB_PROBE_LIB'])
world = int(sys.argv[1]); r = int(sys.argv[2])
case = bench.make_case(bench.WORKLOADS['c2'])
b = shard_bounds(case['grid']['nwave'], world)
m = engine.LBLSpectrum(case, rt_path='transit', wbegin=int(b[r]), wcount=int(b[r + 1] - b[r]))
m.kmax_exchange = lambda t: None
for _ in range(3):
    m.run()
torch.cuda.synchronize()
print('kernel', m.lbl.last_gather_kernel, 'wcount', int(b[r+1]-b[r]))
