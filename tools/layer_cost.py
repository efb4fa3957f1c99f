"""Where the gather's time goes, layer by layer: for layer l of the workload every layer of the
atmosphere is set to l's state, so that one launch does 80x the work of that layer at full
occupancy; prints per-layer gather ms (/nlayers), live records, useful and issued FMA lanes and
the CU-cycles per 256-sample visit.  usage: python tools/layer_cost.py [workload] [stride]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                           # noqa: E402


def main():
    import torch
    from pyratbay_amd import engine, _capi
    if os.environ.get('PB_PROBE_LIB'):          # an instrumented build (tools only)
        _capi.LIBPATH = os.path.abspath(os.environ['PB_PROBE_LIB'])
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    case = bench.make_case(bench.WORKLOADS[name])
    atm, iso = case['atm'], case['iso']
    L = len(atm['temp'])
    model = engine.LBLSpectrum(case, rt_path='transit')
    model.run()
    torch.cuda.synchronize()
    reps = 5
    rows = []
    for l in list(range(0, L, stride)) + [L - 1]:
        temp = np.full(L, atm['temp'][l])
        dens = np.repeat(np.asarray(atm['dens'])[l:l + 1], L, 0)
        isoz = np.repeat(np.asarray(iso['isoz'])[:, l:l + 1], L, 1)
        model.set_atmosphere(temp, dens, isoz)
        model.extinction()
        model.lbl.timing_begin(reps)
        for _ in range(reps):
            model.extinction()
        torch.cuda.synchronize()
        ms, n = model.lbl.timing_end()
        w = model.lbl.last_work() or {}
        ofac, _ = model.lbl.last_state(L, 1)
        ms = ms / max(n, 1)
        visits = w.get('fma_lanes_issued', 0) / 256.0
        cyc = ms * 1e-3 * 2.4e9 * 256 / visits if visits else float('nan')
        rows.append((l, ms / L * 1e3, w.get('live_records', 0) / L,
                     w.get('fma_lanes_useful', 0) / L, w.get('fma_lanes_issued', 0) / L, cyc,
                     int(ofac[0]), model.lbl.last_gather_kernel))
    print('layer  us/layer  live_rec  useful_lanes  issued_lanes  CUcyc/visit  ofactor  kernel')
    for r in rows:
        print(f'{r[0]:5d} {r[1]:9.2f} {r[2]:9.0f} {r[3]:13.3e} {r[4]:13.3e} {r[5]:10.1f} {r[6]:7d}  {r[7]}')


if __name__ == '__main__':
    main()
