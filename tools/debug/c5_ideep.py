import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tools import bench_c5
from pyratbay_amd import engine
inp = bench_c5.inputs()
g, atm = inp['grid'], inp['atm']
nl, nw = atm['nlayers'], g['nwave']
model = engine.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'], atm['rstar'])
def ideeps(seed):
    temps, dens, radius = bench_c5.walkers(inp, 64, seed)
    td, dd, rd = engine.dev(temps), engine.dev(dens), engine.dev(radius)
    ec = engine.interp_ec_batch(model.etable, model.ttable, td, dd)
    path = engine.transit_path_device(rd, 0)
    _, _, ideep = engine.transit_spectrum_batch(ec, path, rd, atm['rstar'], 0, nl, 10.0, want_depth=True)
    return ideep.cpu().numpy()
base = ideeps(700)[0]
order = np.argsort(base, kind='stable')
bs = base[order]
print('base ideep: min', bs.min(), 'max', bs.max(), 'mean', bs.mean())
for seed in (700, 701, 702, 703):
    idp = ideeps(seed)[:, order]          # [64, W] in base order
    d = idp - bs[None]
    print('seed', seed, 'delta ideep: min', d.min(), 'max', d.max(), 'pct', np.percentile(d, [1, 50, 99, 99.9]))
    W32 = (nw // 32) * 32
    need = idp[:, :W32].reshape(64, -1, 32).max(2)          # per (walker, wave of 32 cols)
    basemax = bs[:W32].reshape(-1, 32).max(1)
    tiles_read = (need // 16 + 1)
    print('   transit reads (tile granular): %.3f of ec' % (tiles_read.sum() * 16 * 32 / (64 * nl * W32)))
    for margin in (0, 2, 4, 8, 12, 16):
        # limit per 256-column block: tile containing (block max of base + margin)
        W256 = (nw // 256) * 256
        bmax = bs[:W256].reshape(-1, 256).max(1)
        lim_tile = np.minimum((bmax + margin) // 16, (nl - 1) // 16)       # last tile interpolated
        lim_wave = np.repeat(lim_tile, 8)[:need.shape[1]]
        needw = need[:, :len(lim_wave)] // 16
        over = needw > lim_wave[None]
        frac_written = ((lim_tile + 1) * 16).clip(max=nl).sum() * 256 / (nl * W256)
        print(f'   margin {margin:2d}: ec written {frac_written:.3f}; overrun (walker, wave) {over.mean():.5f}; walkers with any overrun {over.any(1).mean():.2f}')
