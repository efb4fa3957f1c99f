import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from pyratbay_amd import engine
case = bench.make_case(bench.WORKLOADS['c2-res'])
m = engine.LBLSpectrum(case)
print('table bytes before %.3f GB' % (m.voigt.device_bytes / 1e9))
m.run(); torch.cuda.synchronize()
print('table bytes after  %.3f GB' % (m.voigt.device_bytes / 1e9))
free, total = torch.cuda.mem_get_info()
print('device memory in use %.1f GB' % ((total - free) / 1e9))
