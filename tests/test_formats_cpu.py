"""On-disk formats either side of the path: TLI line lists and
.npz opacity tables.  CPU only."""
import struct
import sys

import numpy as np
import pytest

from pyratbay_amd import tli, synth


def _databases(seed=2):
    rng = np.random.default_rng(seed)
    temps = np.linspace(100.0, 3000.0, 7)
    dbs = []
    iso0 = 0
    for name, mol, niso, n in (('HITRAN H2O', 'H2O', 3, 500), ('HITEMP CO', 'CO', 2, 300)):
        counts = rng.multinomial(n, np.ones(niso) / niso)
        wn = np.concatenate([np.sort(rng.uniform(4000, 4100, c)) for c in counts])
        iso_id = np.concatenate([np.full(c, iso0 + j) for j, c in enumerate(counts)])
        dbs.append(dict(name=name, molecule=mol, temperatures=temps,
                        isotopes=[f'{mol}_{j}' for j in range(niso)],
                        iso_mass=18.0 + np.arange(niso), iso_ratio=0.9 / (1 + np.arange(niso)),
                        partition=1 + np.outer(1 + np.arange(niso), temps**1.5) / 10,
                        wn=wn, iso_id=iso_id, elow=rng.uniform(0, 5000, n),
                        gf=10**rng.uniform(-9, -3, n)))
        iso0 += niso
    return dbs


def test_tli_round_trip(tmp_path):
    dbs = _databases()
    path = str(tmp_path / 'mock.tli')
    tli.write_tli(path, dbs)
    # header exactly as lread.py:277-283 writes it
    raw = open(path, 'rb').read(32)
    assert raw[:1].decode() == sys.byteorder[0]
    assert struct.unpack('3h', raw[1:7]) == (6, 5, 0)
    out, wn, gf, elow, iso_id, meta = tli.read_tli(path)
    assert meta['n_lines'] == 800
    assert [d['name'] for d in out] == ['HITRAN H2O', 'HITEMP CO']
    np.testing.assert_array_equal(out[1]['partition'], dbs[1]['partition'])
    np.testing.assert_array_equal(wn, np.concatenate([d['wn'] for d in dbs]))
    np.testing.assert_array_equal(gf, np.concatenate([d['gf'] for d in dbs]))
    np.testing.assert_array_equal(elow, np.concatenate([d['elow'] for d in dbs]))
    np.testing.assert_array_equal(iso_id, np.concatenate([d['iso_id'] for d in dbs]))
    # file size check of the reader (line_by_line.py:392-404)
    with open(path, 'ab') as f:
        f.write(b'\0\0')
    with pytest.raises(ValueError, match='number of transitions'):
        tli.read_tli(path)


def test_tli_range_selection(tmp_path):
    """Per-isotope [wn_low, wn_high] selection, boundaries included
    (line_by_line.py:414-470)."""
    dbs = _databases(seed=9)
    path = str(tmp_path / 'mock.tli')
    tli.write_tli(path, dbs)
    lo, hi = 4020.0, 4060.0
    _, wn, gf, elow, iso_id, _ = tli.read_tli(path, lo, hi)
    all_wn = np.concatenate([d['wn'] for d in dbs])
    all_id = np.concatenate([d['iso_id'] for d in dbs])
    keep = (all_wn >= lo) & (all_wn <= hi)
    np.testing.assert_array_equal(wn, all_wn[keep])
    np.testing.assert_array_equal(iso_id, all_id[keep])
    assert np.all(np.diff(iso_id) >= 0)
    _, wn0, *_ = tli.read_tli(path, 9000.0, 9100.0)
    assert len(wn0) == 0


def test_opacity_file_format(tmp_path):
    from pyratbay_amd import opacity_table as ot
    temp = np.linspace(300, 3000, 4)
    press = np.logspace(-6, 2, 5)
    wn = np.linspace(4000, 4010, 11)
    op = np.random.default_rng(0).uniform(size=(4, 5, 11))
    path = str(tmp_path / 'cs.npz')
    ot.write_opacity(path, 'H2O', temp, press, wn, op)
    with np.load(path, allow_pickle=True) as f:          # keys of io.py:596-604
        assert sorted(f.files) == ['opacity', 'pressure', 'species', 'temperature', 'units',
                                   'wavenumber']
        assert list(f['species']) == ['H2O']
        assert f['units'].item()['cross section'] == 'cm2 molecule-1'
    units, species, t, p, w, o = ot.read_opacity(path)
    assert species == 'H2O' and units['pressure'] == 'bar'
    np.testing.assert_array_equal(o, op)
    np.testing.assert_array_equal(ot.read_opacity(path, 'opacity'), op)
    assert ot.read_opacity(path, 'arrays')[0] == 'H2O'
    with pytest.raises(ValueError):
        ot.write_opacity(path, ['H2O'], temp, press, wn, op)
