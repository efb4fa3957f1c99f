"""Runs one workload through an instrumented library (PB_PROBE_LIB, built by tools/probe_staged.py
with bit 8) so that the workgroup start / end times of the last gather launch land in
$PB_PROBE_TIMES; analyse with tools/wg_times.py.  usage: python tools/wg_probe_run.py [workload]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyratbay_amd import engine, _capi
_capi.LIBPATH = os.path.abspath(os.environ['PB_PROBE_LIB'])
name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
case = bench.make_case(bench.WORKLOADS[name])
m = engine.LBLSpectrum(case, rt_path=bench.WORKLOADS[name].get('rt_path', 'transit'))
for _ in range(3):
    m.run()
torch.cuda.synchronize()
