#!/usr/bin/env python3
"""A/B of the retrieval batch at C5's shape (64 walkers x 80 layers x 1e5 wavenumbers): the
one-pass kernel (pb_table_transit_batch, PB_TABLE_TRANSIT=1) against the two passes over a stored
ec (the default).  ms per 64-walker batch, HIP events."""
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
    model = engine.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'],
                                 atm['rstar'], rt_path='transit')
    bands = engine.PassBands(g['wn'], inp['bands'])
    temps, dens, radius = (engine.dev(a) for a in bench_c5.walkers(inp, 64, 1))
    reps = int(os.environ.get('REPS', '12'))

    def run(label, env):
        os.environ.pop('PB_TABLE_TRANSIT', None)
        os.environ.update(env)
        for _ in range(3):
            out = model.eval_bands(temps, dens, bands, radius=radius)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = model.eval_bands(temps, dens, bands, radius=radius)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f'{label:34s} {ms:7.3f} ms per 64 walkers = {64e3 / ms:8.0f} evals/s', flush=True)
        return out.cpu().numpy()

    ref = run('two passes (stored ec)', {'PB_TABLE_TRANSIT': '0'})
    for pair, tb, pd in (('1', '', ''), ('1', '', '1'), ('0', '', '')):
        os.environ['PB_TP_TB'] = tb
        os.environ['PB_TP_PD'] = pd
        got = run(f"one pass, {'two walkers' if pair == '1' else 'one walker'} per wavefront {tb} {pd}",
                  {'PB_TABLE_TRANSIT': '1', 'PB_TT_PAIR': pair})
        os.environ.pop('PB_TT_PAIR', None)
        os.environ.pop('PB_TP_TB', None)
        os.environ.pop('PB_TP_PD', None)
        err = float(np.max(np.abs(got / ref - 1)))
        assert err <= 1e-13, err
    ref2 = run('two passes (stored ec), again', {'PB_TABLE_TRANSIT': '0'})
    assert np.array_equal(ref, ref2)


if __name__ == '__main__':
    main()
