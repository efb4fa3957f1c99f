#!/usr/bin/env python3
"""Experiment: the optical-depth pass of a 64-walker batch split between the matrix-core kernel
(k_transit_mfma) and the vector kernel (k_transit_pair) running CONCURRENTLY on two streams --
gfx950's FP64 matrix pipe and its vector ALU have the same peak rate and can work side by side.
ms per 64 walkers at C5's shape for every split."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from pyratbay_amd import engine
    from tools import bench_c5
    inp = bench_c5.inputs()
    g, atm = inp['grid'], inp['atm']
    temps, dens, radius = (engine.dev(a) for a in bench_c5.walkers(inp, 64, 1))
    et, tt = engine.dev(inp['etable']), engine.dev(inp['ttable'])
    ec = engine.interp_ec_batch(et, tt, temps, dens)
    path = engine.transit_path_device(radius, 0)
    L = atm['nlayers']
    sa, sb = engine.side_streams(2)
    reps = 10

    def run(n1):
        def once():
            cur = torch.cuda.current_stream()
            outs = []
            if n1 > 0:
                sa.wait_stream(cur)
                with torch.cuda.stream(sa):
                    os.environ['PB_TRANSIT_MFMA'] = '1'
                    outs.append(engine.transit_spectrum_batch(ec[:n1], path[:n1], radius[:n1],
                                                              atm['rstar'], 0, L, 10.0))
                cur.wait_stream(sa)
            if n1 < 64:
                sb.wait_stream(cur)
                with torch.cuda.stream(sb):
                    os.environ['PB_TRANSIT_MFMA'] = '0'
                    outs.append(engine.transit_spectrum_batch(ec[n1:], path[n1:], radius[n1:],
                                                              atm['rstar'], 0, L, 10.0))
                cur.wait_stream(sb)
            return torch.cat(outs)
        for _ in range(2):
            out = once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = once()
        e1.record()
        torch.cuda.synchronize()
        print(f'matrix cores: {n1:2d} walkers, vector ALU: {64 - n1:2d}   '
              f'{e0.elapsed_time(e1) / reps:6.3f} ms', flush=True)
        return out

    ref = run(64)
    for n1 in (0, 24, 32, 36, 40, 44, 48, 56, 64):
        got = run(n1)
        assert float(torch.max(torch.abs(got / ref - 1))) < 1e-12


if __name__ == '__main__':
    main()
