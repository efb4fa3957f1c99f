import numpy as np, sys
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 2).astype(np.int64)
ok = t[:, 1] > 0
t = t[ok]
t0 = t[:, 0].min(); T = t[:, 1].max() - t0
s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0     # us
print(f'workgroups {len(t)}  launch {T/100:.1f} us  mean life {np.mean(e-s):.1f} us  min {np.min(e-s):.1f}  max {np.max(e-s):.1f}')
busy = np.sum(e - s)
print(f'slot-time busy / (1024 x launch) = {busy / (1024 * T / 100.0):.3f}')
for q in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 1.0):
    tt = q * T / 100.0
    act = np.sum((s <= tt) & (e > tt))
    print(f't = {tt:7.1f} us ({q:.2f})  active workgroups {act}')
# start-time distribution: how many start at t=0 (first wave)
print('started within 5 us:', np.sum(s < 5.0), ' end-time percentiles (us):', np.percentile(e, [5, 25, 50, 75, 95, 100]).round(1))
