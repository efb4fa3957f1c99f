#!/usr/bin/env python3
"""Generate tests/golden/g15_line_sample.npz: what the reference's loader of sampled cross
sections (`Line_Sample.__init__`, pyratbay/opacity/line_sampling.py:60-275, with
`tools.interpolate_opacity`, tools/tools.py:1026-1107) builds from several `.npz` opacity files
(build container only; the package is imported as in make_golden_e2e.py):

    python tests/golden/make_golden_line_sample.py

Four small synthetic files (written in the reference's own format with ITS writer): H2O on one
(T, p) grid, CO on a finer pressure grid and other temperatures, a second H2O file that must be
ADDED to the first, and CH4 on the target grid itself.  Cases: the files' own grid; a new pressure
grid (log-linear interpolation, constant beyond the table); new pressures and temperatures; a
wavenumber window with thinning.  Stored: the files' arrays (inputs) and, per case, the species
order, wavenumbers and the cross-section table the reference ends up with -- data only."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e          # noqa: E402


def main():
    work = tempfile.mkdtemp(prefix='pb_ls_')
    try:
        pb = e2e.reference_package(work)
        import pyratbay.io as io
        import pyratbay.opacity as op
        rng = np.random.default_rng(15)
        wn = 4000.0 + 0.5 * np.arange(161)
        files = {
            'h2o_a': dict(species='H2O', temp=np.array([500.0, 1000.0, 1500.0, 2000.0]),
                          press=np.logspace(-5, 1, 7)),
            'co': dict(species='CO', temp=np.array([400.0, 900.0, 1400.0, 1900.0, 2400.0]),
                       press=np.logspace(-6, 2, 17)),
            'h2o_b': dict(species='H2O', temp=np.array([500.0, 1000.0, 1500.0, 2000.0]),
                          press=np.logspace(-5, 1, 7)),
            'ch4': dict(species='CH4', temp=np.array([500.0, 1000.0, 1500.0, 2000.0]),
                        press=np.logspace(-5, 1, 7)),
        }
        store = {'wn': wn}
        paths = []
        for name, f in files.items():
            nt, nl = len(f['temp']), len(f['press'])
            base = rng.uniform(-28.0, -19.0, (1, 1, len(wn)))
            cs = 10.0**(base + 0.4 * np.log10(f['press'])[None, :, None]
                        + 1.5 * (f['temp'] / 2000.0)[:, None, None]
                        + 0.2 * rng.standard_normal((nt, nl, len(wn))))
            if name == 'h2o_b':
                cs[:, :, 40:60] = 0.0                  # exact zeros: the -230 cap of the log
            path = os.path.join(work, f'cross_section_{name}.npz')
            io.write_opacity(path, f['species'], f['temp'], f['press'], wn, cs)
            paths.append(path)
            store[f'{name}_species'] = f['species']
            store[f'{name}_temp'] = f['temp']
            store[f'{name}_press'] = f['press']
            store[f'{name}_cs'] = cs
        store['file_order'] = np.array(list(files))
        cases = {
            'native': dict(files=[0, 2, 3]),
            'newp': dict(files=[0, 1, 2, 3], pressure=np.logspace(-7, 0.9, 23)),
            'newpt': dict(files=[0, 1, 2, 3], pressure=np.logspace(-6, 1, 15),
                          temperature=np.array([450.0, 700.0, 1000.0, 1750.0, 2300.0])),
            'window': dict(files=[1, 3], pressure=np.logspace(-5, 1, 7),
                           temperature=np.array([500.0, 1000.0, 1500.0, 2000.0]),
                           min_wn=4010.2, max_wn=4060.0, wl_thinning=3),
        }
        for cname, c in cases.items():
            kw = {k: v for k, v in c.items() if k != 'files'}
            ls = op.Line_Sample([paths[i] for i in c['files']], **kw)
            store[f'{cname}_files'] = np.array(c['files'])
            for k, v in kw.items():
                store[f'{cname}_arg_{k}'] = v
            store[f'{cname}_species'] = np.array(ls.species)
            store[f'{cname}_wn'] = ls.wn
            store[f'{cname}_temp'] = ls.temp
            store[f'{cname}_press'] = ls.press
            store[f'{cname}_cs_table'] = ls.cs_table
            print(cname, list(ls.species), ls.cs_table.shape, float(ls.cs_table.min()),
                  float(ls.cs_table.max()))
        # the error the loader raises for a pressure profile beyond the table
        try:
            op.Line_Sample([paths[0]], pressure=np.logspace(-5, 1.5, 7))
            store['beyond_table_error'] = ''
        except ValueError as e:
            store['beyond_table_error'] = str(e)
        np.savez_compressed(os.path.join(HERE, 'g15_line_sample.npz'), **store)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print('g15_line_sample.npz', os.path.getsize(os.path.join(HERE, 'g15_line_sample.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
