"""Analysis of the workgroup timing probe (tools/probe_staged.py bits 8/9, written to
$PB_PROBE_TIMES by an instrumented library): occupancy of the 1024 workgroup slots over the
launch, lifetimes, per-XCD balance and -- with the phase timers -- where a workgroup's time goes.
usage: python tools/wg_times.py file.bin [columns (2 or 6)] [ntiles nlayers nsplit]"""
import sys

import numpy as np

f = sys.argv[1]
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t = np.fromfile(f, dtype=np.uint64).reshape(-1, cols).astype(np.int64)
ids = np.arange(len(t))
ok = t[:, 1] > 0
t0 = t[ok, 0].min()
T = (t[ok, 1].max() - t0) / 100.0
s = (t[:, 0] - t0) / 100.0
e = (t[:, 1] - t0) / 100.0
life = e - s
print(f'workgroups {ok.sum()}  launch {T:.1f} us  mean life {life[ok].mean():.1f} us  '
      f'min {life[ok].min():.1f}  max {life[ok].max():.1f}')
print(f'slot-time busy / (1024 x launch) = {life[ok].sum() / (1024 * T):.3f}')
for q in (0.5, 0.7, 0.8, 0.9, 0.95):
    tt = q * T
    print(f't = {tt:8.1f} us ({q:.2f})  active workgroups {np.sum(ok & (s <= tt) & (e > tt))}')
x = ids & 7
print('per XCD busy slot-time / 128 (us):', [int(life[ok & (x == i)].sum() / 128) for i in range(8)])
print('per XCD last end (us):            ', [int(e[ok & (x == i)].max()) for i in range(8)])
if cols >= 6:
    tot = life[ok].sum()
    for name, c in (('candidate search', 2), ('batch set-up', 3), ('segment steps', 4)):
        print(f'{name:18s} {t[ok, c].sum() / 100.0 / tot:6.1%} of the slot-time')
    print(f'batches per workgroup {t[ok, 5].mean():.1f}; set-up per batch '
          f'{t[ok, 3].sum() / 100.0 / max(t[ok, 5].sum(), 1):.2f} us; search per workgroup '
          f'{t[ok, 2].mean() / 100.0:.1f} us')
if len(sys.argv) > 5:
    ntiles, L, nsplit = (int(v) for v in sys.argv[3:6])
    k = ids >> 3
    grp = k // ntiles
    unit = grp * 8 + np.where(grp & 1, 7 - (ids & 7), ids & 7)
    layer = L - 1 - unit // nsplit
    print('layer  mean life  max life  mean start' + ('  search  set-up  steps (us)' if cols >= 6 else ''))
    for l in range(L - 1, -1, -max(1, L // 10)):
        m = ok & (layer == l)
        if not m.any():
            continue
        extra = ''
        if cols >= 6:
            extra = ''.join(f' {t[m, c].mean() / 100.0:7.1f}' for c in (2, 3, 4))
        print(f'{l:5d} {life[m].mean():10.1f} {life[m].max():9.1f} {s[m].mean():11.1f}{extra}')
