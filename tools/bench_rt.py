"""Device time of the radiative-transfer column kernels at the C2 shape (80 layers x 100 001
samples): plane-parallel depth, emission flux (5-point quadrature), two-stream fluxes, band
integration.  usage: python tools/bench_rt.py [nlayers] [nwave]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyratbay_amd import engine

L = int(sys.argv[1]) if len(sys.argv) > 1 else 80
W = int(sys.argv[2]) if len(sys.argv) > 2 else 100001
engine.require_gpu()
rng = np.random.default_rng(3)
wn = engine.dev(np.linspace(4000.0, 9000.0, W))
temp = engine.dev(np.linspace(1000.0, 1700.0, L))
radius = np.linspace(7.5e9, 7.0e9, L)
intervals = engine.dev(-np.diff(radius))
press = np.logspace(-6, 2, L)
ec = engine.dev(10.0**rng.uniform(-12, -9, (L, W)) * (press[:, None] / 1e-3)**0.8)
mu, wq = engine.default_quadrature()
mu, wq = engine.dev(mu), engine.dev(wq)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'{name:34s} {ms:7.3f} ms   {nbytes / ms / 1e6:7.0f} GB/s on {nbytes / 1e6:.0f} MB')


depth, ideep = engine.plane_parallel_optical_depth(ec, intervals, 0, L, 10.0)
timed('plane_parallel_optical_depth', lambda: engine.plane_parallel_optical_depth(
    ec, intervals, 0, L, 10.0), 16.0 * L * W)
timed('emission_flux (5 mu)', lambda: engine.emission_flux(depth, ideep, wn, temp, mu, wq, 0),
      8.0 * L * W + 8.0 * W)
depth_inf, _ = engine.plane_parallel_optical_depth(ec, intervals, 0, L, np.inf)
f_int = engine.internal_flux(wn, 100.0)
timed('two_stream', lambda: engine.two_stream(depth_inf, wn, temp, f_int, None, 0),
      8.0 * L * W + 16.0 * L * W)
