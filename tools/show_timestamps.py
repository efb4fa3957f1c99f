"""The reference's `pyrat.timestamps` on the device path: stage seconds of one run() per workload
(HIP events, engine.StageTimer), printed like pyrat_obj.py:209-214 prints them.  Under
`rocprofv3 --marker-trace --kernel-trace --stats` the same stages and the collectives appear as
rocTX ranges.  usage: python tools/show_timestamps.py [workload ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine, _capi

print('rocTX marker library found:', bool(_capi.call('pb_roctx_available')))
for name in sys.argv[1:] or ['c2']:
    w = bench.WORKLOADS[name]
    model = engine.LBLSpectrum(bench.make_case(w), rt_path=w.get('rt_path', 'transit'))
    for _ in range(3):
        model.run()
    torch.cuda.synchronize()
    with engine.profiler_range(f'{name}: ten spectra'):
        for _ in range(10):
            model.run()
        ts = model.timestamps
    print(f'{name}: Timestamps (s):\n' + '\n'.join(f'{k:10s}: {v:10.6f}' for k, v in ts.items()))
