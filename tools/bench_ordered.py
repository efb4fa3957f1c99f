#!/usr/bin/env python3
"""The transit stage of the retrieval batch at C5's shape (64 walkers x 80 layers x 1e5
wavenumbers), columns in grid order against columns ordered by the row at which a base model
crosses maxdepth: k_transit_mfma (PB_TRANSIT_MFMA=4), k_transit_mfma_rows in grid order,
pb_transit_spectrum_ordered.  ms per 64-walker launch, HIP events; spectra compared bit for bit."""
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
    L, W = atm['nlayers'], g['nwave']
    model = engine.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'],
                                 atm['rstar'], rt_path='transit')
    temps, dens, radius = (engine.dev(a) for a in bench_c5.walkers(inp, 64, 1))
    reps = int(os.environ.get('REPS', '10'))
    ec = engine.interp_ec_batch(model.etable, model.ttable, temps, dens)
    path = engine.transit_path_device(radius, 0)
    # the order: the base model's first crossing per column
    bt, bd = engine.dev(atm['temp'][None]), engine.dev(atm['dens'][None, :, 2:2 + bench_c5.NSPEC])
    bec = engine.interp_ec_batch(model.etable, model.ttable, bt, bd)
    brad = engine.dev(atm['radius'][None])
    _, _, ideep = engine.transit_spectrum_batch(bec, engine.transit_path_device(brad, 0), brad,
                                                atm['rstar'], 0, L, 10.0, want_depth=True)
    order = torch.sort(ideep[0], stable=True).indices.to(torch.int32)
    ec_ord = ec[:, :, order.long()].contiguous()

    def timed(label, fn):
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        print(f'{label:46s} {e0.elapsed_time(e1) / reps:7.3f} ms', flush=True)
        return out

    os.environ['PB_TRANSIT_MFMA'] = '4'
    ref = timed('k_transit_mfma, grid order', lambda: engine.transit_spectrum_batch(
        ec, path, radius, atm['rstar'], 0, L, 10.0))
    os.environ.pop('PB_TRANSIT_MFMA', None)
    rows = timed('k_transit_mfma_rows, grid order', lambda: engine.transit_spectrum_batch(
        ec, path, radius, atm['rstar'], 0, L, 10.0))
    os.environ.pop('PB_TRANSIT_MFMA', None)
    ordered = timed('k_transit_mfma_rows, ordered columns', lambda: engine.transit_spectrum_ordered(
        ec_ord, path, radius, order, atm['rstar'], 0, L, 10.0))
    print('rows == mfma:', bool(torch.equal(rows, ref)), ' ordered == mfma:',
          bool(torch.equal(ordered, ref)),
          ' max rel', float(((ordered - ref).abs() / ref.abs()).max()))


if __name__ == '__main__':
    main()
