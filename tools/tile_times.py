#!/usr/bin/env python3
"""Per-tile time spread of the extinction gather: every 4096-sample tile
of a workload computed ALONE, as a wavenumber-shard call of one tile over all layers (HIP events,
ms), under three policies of the product library -- the staged kernel with the launch-wide split
only (PB_TILE_SPLIT=0), with the per-tile split (PB_TILE_GLOBAL=0), and the default (per-tile split
+ sparse tiles to the global gather) -- beside the tile's number of groups.  The sum over the tiles
is NOT the time of the whole launch (tiles of a launch share the chip); the spread is the point.
usage: python tools/tile_times.py [workload] [reps]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TILE = 4096


def worker(name, reps):
    import torch
    import bench
    from pyratbay_amd import engine
    case = bench.make_case(bench.WORKLOADS[name])
    m = engine.LBLSpectrum(case)
    nwave = m.nwave
    ntiles = -(-nwave // TILE)
    iown = np.asarray(case['lines']['lwn'])
    wn = np.asarray(case['grid']['wn'])
    edges = np.append(wn[::TILE], wn[-1] + 1.0)
    groups = np.histogram(iown, edges)[0]
    # whole launch first (the per-tile split is sized from the previous call's counts)
    for _ in range(3):
        m.lbl.extinction(m.temp, m.dens, m.isoz)
    out = []
    for t in range(ntiles):
        w0, wc = t * TILE, min(TILE, nwave - t * TILE)
        for _ in range(2):
            m.lbl.extinction(m.temp, m.dens, m.isoz, wbegin=w0, wcount=wc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m.lbl.extinction(m.temp, m.dens, m.isoz, wbegin=w0, wcount=wc)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    print('TILES ' + ' '.join(f'{int(g)}:{ms:.4f}' for g, ms in zip(groups, out)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        worker(sys.argv[2], int(sys.argv[3]))
        return
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2-bands'
    reps = sys.argv[2] if len(sys.argv) > 2 else '5'
    res = {}
    for label, env in (('launch-wide split', {'PB_TILE_SPLIT': '0'}),
                       ('per-tile split', {'PB_TILE_GLOBAL': '0'}),
                       ('per-tile split + sparse tiles global (default)', {})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', name, reps],
                           env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith('TILES ')]
        if not line:
            print(p.stdout[-2000:], p.stderr[-2000:])
            sys.exit(1)
        res[label] = [(int(a.split(':')[0]), float(a.split(':')[1])) for a in line[0].split()[1:]]
    labels = list(res)
    print(f'{name}: one tile of {TILE} samples per call, all layers, ms per call')
    print('tile  lines-in-tile  ' + '  '.join(f'{l[:28]:>28s}' for l in labels))
    nt = len(res[labels[0]])
    for t in range(nt):
        print(f'{t:4d}  {res[labels[0]][t][0]:13d}  ' +
              '  '.join(f'{res[l][t][1]:28.4f}' for l in labels))
    for l in labels:
        ms = np.array([x[1] for x in res[l]])
        print(f'{l}: min {ms.min():.3f}  median {np.median(ms):.3f}  max {ms.max():.3f}  '
              f'max/min {ms.max() / ms.min():.1f}  sum {ms.sum():.2f} ms')


if __name__ == '__main__':
    main()
