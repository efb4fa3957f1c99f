#!/usr/bin/env python3
"""One-off sweep (build container only, needs /root/reference): pyratbay_amd.tli.read_tli against
the reference's read_tli_file (pyrat/line_by_line.py:298-482) on random [wn_low, wn_high] windows,
for the two reference-written fixture files and for random multi-database files written by
pyratbay_amd.tli.write_tli (duplicates, short isotopes, windows on exact line positions).
Result of round 2: 1 500 cases, 0 mismatches for isotopes of >= 3 lines; with 1-2 line isotopes the
reference's reader itself raises (ValueError in its offset arithmetic) or returns lines outside
the window, and the two readers are not compared there (read_tli(strict=True) is the correct one).

    python tests/golden/fuzz_tli_reader.py [cases]
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden_e2e import reference_package, REF           # noqa: E402


def random_file(rng, path):
    from pyratbay_amd import tli
    temps = np.linspace(100.0, 3000.0, int(rng.integers(2, 6)))
    dbs, iso0 = [], 0
    for d in range(int(rng.integers(1, 4))):
        niso = int(rng.integers(1, 5))
        counts = rng.integers(3, int(rng.choice([6, 20, 200])), niso)   # (1-2 line isotopes: the reference reader itself raises)
        wn, ids = [], []
        for j, c in enumerate(counts):
            w = np.sort(rng.uniform(4000, 4100, c))
            if c > 3 and rng.random() < 0.5:
                w[1:3] = w[0]                               # duplicates at the low end
                w = np.sort(w)
            wn.append(w)
            ids.append(np.full(c, iso0 + j))
        n = int(np.sum(counts))
        dbs.append(dict(name=f'db{d}', molecule=f'M{d}', temperatures=temps,
                        isotopes=[f'{d}{j}' for j in range(niso)],
                        iso_mass=10.0 + np.arange(niso), iso_ratio=np.ones(niso) / niso,
                        partition=1 + np.outer(1 + np.arange(niso), temps),
                        wn=np.concatenate(wn), iso_id=np.concatenate(ids),
                        elow=rng.uniform(0, 5000, n), gf=10**rng.uniform(-9, -3, n)))
        iso0 += niso
    tli.write_tli(path, dbs)
    return np.concatenate([d['wn'] for d in dbs])


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_tlifuzz_')
    bad = 0
    try:
        reference_package(work)
        import mc3.utils as mu
        from pyratbay.pyrat.line_by_line import read_tli_file
        from pyratbay_amd import tli
        rng = np.random.default_rng(31)
        files = [(os.path.join(HERE, 'g13_mock_h2o.tli'), None), (os.path.join(HERE, 'g13_two_db.tli'), None)]
        for k in range(ncases):
            if k % 3 == 0 or len(files) < 3:
                p = os.path.join(work, f'r{k}.tli')
                files.append((p, random_file(rng, p)))
            path, allwn = files[int(rng.integers(0, len(files)))]
            if allwn is None:
                allwn = tli.read_tli(path, strict=True)[1]
            lo, hi = float(allwn.min()), float(allwn.max())
            pick = lambda: float(rng.choice(allwn)) if rng.random() < 0.4 else float(rng.uniform(lo - 2, hi + 2))
            a, b = sorted((pick(), pick()))
            if rng.random() < 0.1:
                a = -np.inf
            if rng.random() < 0.1:
                b = np.inf
            try:
                _, w, g, e, i_ = read_tli_file(path, a, b, mu.Log(verb=0))
            except Exception as ex:                        # the reference raises on some inputs
                print(f'case {k}: reference raised {type(ex).__name__}: {ex} on [{a}, {b}] -- skipped')
                continue
            _, w2, g2, e2, i2, _ = tli.read_tli(path, a, b)
            same = (np.array_equal(w, w2) and np.array_equal(g, g2) and np.array_equal(e, e2)
                    and np.array_equal(i_, i2))
            if not same:
                bad += 1
                print(f'case {k}: MISMATCH file {os.path.basename(path)} window [{a!r}, {b!r}]: '
                      f'reference {len(w)} lines, ours {len(w2)}')
                meta = tli.read_tli(path, strict=True)
                per = meta[5]['lines_per_isotope']
                full = meta[1]
                o = 0
                for j, c in enumerate(per):
                    print(f'   isotope {j}: {c} lines {full[o:o + c][:6]}{" ..." if c > 6 else ""} .. {full[o + c - 1]}')
                    o += c
                print('   reference:', w, i_)
                print('   ours     :', w2, i2)
        print(f'{ncases} cases, {bad} mismatches')
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
