#!/usr/bin/env python3
"""Fixture G13: TLI files written by the REFERENCE's own writer (pyratbay/opacity/lread.py:277-314,
`runmode = tli`) and what the reference's reader (pyrat/line_by_line.py:298-482) returns for them.

Build container only (needs /root/reference); same scratch import as make_golden_e2e.py.

    python tests/golden/make_golden_tli.py

Writes
  tests/golden/g13_mock_h2o.tli      the bundled mock HITRAN H2O list (888 lines, 4 isotopes)
  tests/golden/g13_two_db.tli        two databases in one file: that H2O list + the bundled
                                     mock HITRAN CO2 list (tests/inputs/mock_02_hit12.tar.gz)
  tests/golden/g13_tli_reader.npz    for each file and several [wn_low, wn_high] windows (full
                                     range, interior, on a line, between two lines, below / above
                                     an isotope's lines, nothing selected): the arrays
                                     read_tli_file() returns, plus the Database headers
Only data: bytes the reference wrote and arrays it returned.
"""
import os
import shutil
import sys
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_e2e import reference_package, REF, CFG_TLI      # noqa: E402

CFG_TWO = '''
[pyrat]
runmode = tli
logfile = {work}/two_db.log
tlifile = {work}/two_db.tli
dblist = {ref}/tests/inputs/Mock_HITRAN_H2O_1.00-1.01um.par
    {work}/mock_02_hit12.par
dbtype = hitran hitran
pflist = tips tips
wl_low = 1.00 um
wl_high = 1.52 um
verb = 0
'''


def windows(wn):
    """Range-selection cases for a sorted-per-isotope wavenumber column."""
    lo, hi = float(wn.min()), float(wn.max())
    mid = float(np.sort(wn)[len(wn) // 2])
    s = np.unique(wn)
    gap_lo, gap_hi = s[len(s) // 3], s[len(s) // 3 + 1]
    return [
        (-np.inf, np.inf), (lo, hi), (lo - 5.0, hi + 5.0),
        (lo + 0.3 * (hi - lo), lo + 0.6 * (hi - lo)),
        (mid, mid),                                  # exactly one value (and its duplicates)
        (mid, hi + 1.0), (lo - 1.0, mid),
        (gap_lo + 0.25 * (gap_hi - gap_lo), gap_lo + 0.75 * (gap_hi - gap_lo)),   # between lines
        (hi + 1.0, hi + 2.0), (lo - 2.0, lo - 1.0),  # nothing
    ]


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_tli_')
    try:
        pb = reference_package(work)
        import mc3.utils as mu
        from pyratbay.pyrat.line_by_line import read_tli_file
        with tarfile.open(os.path.join(REF, 'tests/inputs/mock_02_hit12.tar.gz')) as tar:
            member = [m for m in tar.getmembers() if m.name.endswith('mock_02_hit12.par')
                      and not os.path.basename(m.name).startswith('._')][0]
            with open(os.path.join(work, 'mock_02_hit12.par'), 'wb') as f:
                f.write(tar.extractfile(member).read())
        store = {}
        for name, cfg_text in (('mock_h2o', CFG_TLI), ('two_db', CFG_TWO)):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(cfg_text.format(work=work, ref=REF))
            pb.run(cfg)
            src = os.path.join(work, name + '.tli')
            shutil.copy(src, os.path.join(HERE, f'g13_{name}.tli'))
            log = mu.Log(verb=0)
            dbs, wn, gf, elow, iso = read_tli_file(src, -np.inf, np.inf, log)
            store[f'{name}_ndb'] = len(dbs)
            for i, db in enumerate(dbs):
                store[f'{name}_db{i}_name'] = db.name
                store[f'{name}_db{i}_molname'] = db.molname
                store[f'{name}_db{i}_niso'] = db.niso
                store[f'{name}_db{i}_temp'] = db.temp
                store[f'{name}_db{i}_iso_pf'] = db.iso_pf
                store[f'{name}_db{i}_iso_name'] = db.iso_name
                store[f'{name}_db{i}_iso_mass'] = db.iso_mass
                store[f'{name}_db{i}_iso_ratio'] = db.iso_ratio
            wins = windows(wn)
            store[f'{name}_windows'] = np.array(wins)
            for k, (a, b) in enumerate(wins):
                dbs, w, g, e, i_ = read_tli_file(src, a, b, mu.Log(verb=0))
                store[f'{name}_w{k}_wn'] = w
                store[f'{name}_w{k}_gf'] = g
                store[f'{name}_w{k}_elow'] = e
                store[f'{name}_w{k}_iso'] = i_
            print(name, 'lines', len(wn), 'databases', len(dbs),
                  'bytes', os.path.getsize(src), 'iso ids', np.unique(iso))
        np.savez_compressed(os.path.join(HERE, 'g13_tli_reader.npz'), **store)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
