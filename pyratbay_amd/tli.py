"""TLI (transition line information) files: the reference's binary line-list format.

Writer layout: pyratbay/opacity/lread.py:277-314; reader semantics:
pyratbay/pyrat/line_by_line.py:298-482 (+ Database header :485-540).  Native endianness,
version 6.5.0:

    1s endian ('l'|'b') . 3h version . 2d wn_min wn_max . h n_databases
    per database: h+name . h+molecule . hh ntemp niso . ntemp d temperatures
                  per isotope: h+name . d mass . d ratio . ntemp d partition function
    i n_lines . i n_iso_total . n_iso_total i lines-per-isotope
    four contiguous column blocks over all lines, sorted by (isotope, wavenumber):
        n_lines d wavenumber . n_lines h isotope index . n_lines d Elow . n_lines d gf

The four column blocks are read with one `np.fromfile` each per isotope range, which is
also the layout the device wants (SoA), so ingestion is a straight copy.
"""
import struct
import sys

import numpy as np

VERSION = (6, 5, 0)


def _pack_str(f, s):
    b = s.encode('utf-8')
    f.write(struct.pack(f'h{len(b)}s', len(b), b))


def _read_str(f):
    n, = struct.unpack('h', f.read(2))
    return f.read(n).decode('utf-8')


def write_tli(path, databases, wn_min=None, wn_max=None, version=VERSION):
    """databases: list of dicts with keys name, molecule, temperatures[ntemp],
    isotopes (names), iso_mass, iso_ratio, partition[niso, ntemp], and the line arrays
    wn, iso_id, elow, gf sorted by (isotope, wavenumber) within the database.  iso_id counts
    over the concatenated isotope list of all databases; the FILE holds it relative to the
    line's own database, as lread.py:181-209,309 writes it."""
    all_wn = np.concatenate([np.asarray(d['wn'], float) for d in databases])
    wn_min = float(all_wn.min()) if wn_min is None else wn_min
    wn_max = float(all_wn.max()) if wn_max is None else wn_max
    with open(path, 'wb') as f:
        f.write(struct.pack('s', sys.byteorder[0].encode()))
        f.write(struct.pack('3h', *version))
        f.write(struct.pack('2d', wn_min, wn_max))
        f.write(struct.pack('h', len(databases)))
        n_iso_lines = []
        iso_offset = 0
        for d in databases:
            _pack_str(f, d['name'])
            _pack_str(f, d['molecule'])
            temps = np.asarray(d['temperatures'], float)
            niso = len(d['isotopes'])
            f.write(struct.pack('hh', len(temps), niso))
            f.write(temps.tobytes())
            pf = np.asarray(d['partition'], float)
            for j in range(niso):
                _pack_str(f, str(d['isotopes'][j]))
                f.write(struct.pack('d', float(d['iso_mass'][j])))
                f.write(struct.pack('d', float(d['iso_ratio'][j])))
                f.write(pf[j].tobytes())
            ids = np.asarray(d['iso_id'])
            for j in range(niso):
                n_iso_lines.append(int(np.sum(ids == iso_offset + j)))
            iso_offset += niso
        f.write(struct.pack('i', len(all_wn)))
        f.write(struct.pack('i', len(n_iso_lines)))
        f.write(np.asarray(n_iso_lines, np.int32).tobytes())
        f.write(all_wn.tobytes())
        first, stored = 0, []
        for d in databases:
            stored.append(np.asarray(d['iso_id'], np.int64) - first)
            first += len(d['isotopes'])
        f.write(np.concatenate(stored).astype(np.int16).tobytes())
        f.write(np.concatenate([np.asarray(d['elow'], float) for d in databases]).tobytes())
        f.write(np.concatenate([np.asarray(d['gf'], float) for d in databases]).tobytes())


def read_tli(path, wn_low=-np.inf, wn_high=np.inf, strict=False):
    """Returns (databases, wn, gf, elow, iso_id, meta): the lines restricted per isotope to
    wn_low <= wn <= wn_high, like read_tli_file (line_by_line.py:298-482; its range search
    tools.py:219-311 keeps duplicates of a boundary value), + the file header.  The four
    column blocks are memory-mapped: only the selected ranges are read from disk.

    One quirk of the reference is reproduced unless strict=True: when the window lies entirely
    ABOVE the lines of an isotope that is not the first in the file, the reference's
    "not found" index -1 has the isotope's offset added before it is tested
    (line_by_line.py:418-424), so it returns the last line of the previous isotope followed
    by ALL lines of that isotope.  (Harmless downstream: the extinction loop skips lines
    outside the spectral grid, _extcoeff.c:234-236.)  An isotope without lines makes the
    reference raise (tools.py:264); here it is skipped."""
    with open(path, 'rb') as f:
        endian = f.read(1).decode()
        if endian != sys.byteorder[0]:
            raise ValueError(f'Incompatible endianness between TLI file ({endian}) and '
                             f'this machine ({sys.byteorder[0]})')
        ver = struct.unpack('3h', f.read(6))
        if ver[0] != 6 or ver[1] not in (1, 2, 3, 4, 5):
            raise ValueError('Incompatible TLI version.  The TLI file must be created '
                             'with Lineread version 6.1-6.5.')
        file_wn_min, file_wn_max = struct.unpack('2d', f.read(16))
        ndb, = struct.unpack('h', f.read(2))
        databases = []
        for _ in range(ndb):
            name = _read_str(f)
            molecule = _read_str(f)
            ntemp, niso = struct.unpack('hh', f.read(4))
            temps = np.frombuffer(f.read(8 * ntemp), float).copy()
            isotopes, mass, ratio, pf = [], [], [], []
            for _ in range(niso):
                isotopes.append(_read_str(f))
                m, r = struct.unpack('2d', f.read(16))
                mass.append(m)
                ratio.append(r)
                pf.append(np.frombuffer(f.read(8 * ntemp), float).copy())
            databases.append(dict(name=name, molecule=molecule, temperatures=temps,
                                  isotopes=isotopes, iso_mass=np.array(mass),
                                  iso_ratio=np.array(ratio), partition=np.array(pf)))
        n_lines, = struct.unpack('i', f.read(4))
        n_iso, = struct.unpack('i', f.read(4))
        per_iso = np.frombuffer(f.read(4 * n_iso), np.int32).copy()
        start = f.tell()
        f.seek(0, 2)
        if (f.tell() - start) != n_lines * 26:
            raise ValueError('The remaining data file size does not correspond to the '
                             f'number of transitions ({n_lines})')
    off_wn = start
    off_iso = off_wn + 8 * n_lines
    off_el = off_iso + 2 * n_lines
    off_gf = off_el + 8 * n_lines
    wn_all = np.memmap(path, np.float64, 'r', off_wn, (n_lines,))
    pieces = []
    lo = 0
    for n in per_iso:
        n = int(n)
        if n == 0:
            continue
        seg = wn_all[lo:lo + n]
        a = lo + int(np.searchsorted(seg, wn_low, 'left'))
        b = lo + int(np.searchsorted(seg, wn_high, 'right'))
        if not strict and lo > 0 and seg[-1] < wn_low and wn_high >= seg[0]:
            a, b = lo - 1, lo + n                 # the reference's off-by-offset (see above)
        if b > a:
            pieces.append((a, b))
        lo += n

    def gather(offset, dtype):
        col = np.memmap(path, dtype, 'r', offset, (n_lines,))
        if not pieces:
            return np.zeros(0, dtype)
        return np.concatenate([np.asarray(col[a:b]) for a, b in pieces])

    wn = gather(off_wn, np.float64)
    iso_id = gather(off_iso, np.int16)
    elow = gather(off_el, np.float64)
    gf = gather(off_gf, np.float64)
    # index of every returned line's isotope in the concatenated isotope list of the file (the
    # stored column restarts at 0 in every database: lread.py:181-209)
    bounds = np.concatenate([[0], np.cumsum(per_iso)])
    rows = np.concatenate([np.arange(a, b) for a, b in pieces]) if pieces else np.zeros(0, int)
    iso_global = (np.searchsorted(bounds, rows, 'right') - 1).astype(np.int32)
    meta = dict(wn_min=file_wn_min, wn_max=file_wn_max, version=ver, n_lines=n_lines,
                lines_per_isotope=per_iso, iso_global=iso_global)
    return databases, wn, gf, elow, iso_id, meta


def _slinear(ttab, pf, temperature):
    """scipy.interpolate.interp1d(ttab, pf, kind='slinear')(temperature) restated: the
    first-order B-spline as SciPy's de Boor recurrence evaluates it (w = 1/(t_hi - t_lo), basis
    w (t_hi - T) and w (T - t_lo), interval t_lo <= T < t_hi with the last one closed) -- bit
    for bit -- and interp1d's error for a temperature outside the table."""
    ttab = np.asarray(ttab, float)
    pf = np.atleast_2d(np.asarray(pf, float))
    x = np.asarray(temperature, float)
    if ttab.size < 2:
        raise ValueError('a partition-function table needs at least two temperatures')
    if np.any(x < ttab[0]):
        raise ValueError("A value in the temperature profile is below the partition-function "
                         f"table's minimum temperature ({ttab[0]} K)")
    if np.any(x > ttab[-1]):
        raise ValueError("A value in the temperature profile is above the partition-function "
                         f"table's maximum temperature ({ttab[-1]} K)")
    lo = np.clip(np.searchsorted(ttab, x, 'right') - 1, 0, ttab.size - 2)
    xa, xb = ttab[lo], ttab[lo + 1]
    w = 1.0 / (xb - xa)
    return pf[:, lo] * (w * (xb - x)) + pf[:, lo + 1] * (w * (x - xa))


def iso_partition(databases, temperature):
    """Partition function of every isotope of the file (concatenated over its databases, the
    order of read_tli's iso_id) at the layer temperatures: [niso_total, nlayers].

    Reference: Line_By_Line.__init__ builds scipy.interpolate.interp1d(db.temp, db.iso_pf[j],
    kind='slinear') per isotope (pyratbay/pyrat/line_by_line.py:156-158) and
    calc_extinction_coefficient evaluates them at the temperature profile on EVERY call
    (:219-222) -- an LBL retrieval changes T each time.  Piecewise linear in T per isotope;
    a temperature outside a database's table raises ValueError like interp1d does."""
    return np.concatenate([_slinear(db['temperatures'], db['partition'], temperature)
                           for db in databases], axis=0)
