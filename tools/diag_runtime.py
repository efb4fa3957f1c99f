"""Diagnostic: which HIP runtime(s) does a process hold after torch + libpbhip load?"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import torch
print('torch', torch.__version__, 'avail', torch.cuda.is_available())
from pyratbay_amd import _capi
lib = _capi.lib()
maps = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l})
print('\n'.join(maps))
n = C.c_int(-1)
rc = lib.pb_device_count(C.byref(n))
print('pb_device_count rc', rc, 'n', n.value, lib.pb_last_error())
x = torch.zeros(4, device='cuda')
rc = lib.pb_device_count(C.byref(n))
print('after torch alloc: rc', rc, 'n', n.value)
