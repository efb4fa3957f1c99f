#!/usr/bin/env python3
"""Fixture G16: the REAL reference package end to end on a multi-database line list in every
spectral-sampling mode (what G6 does not pin -- `resolution` and `wlstep`
modes, several species / databases in one run, `quadrature = n`).

Build container only (needs /root/reference); same scratch import as make_golden_e2e.py:

    python tests/golden/make_golden_multi.py

  * TLI files (`runmode = tli`) from every mock HITRAN / HITEMP list the reference bundles, one
    file per molecule as the reference's own multi-species configurations do
    (tests/configs/tli_multiple_opacity_*.cfg):
      h2o   tests/inputs/Mock_HITRAN_H2O_1.00-1.01um.par       4 isotopes,  9900-10000 cm-1
            (= tests/golden/g13_mock_h2o.tli, checked byte for byte, not stored again)
      co2   mock_02_hit12 + mock_02_3750-4000_HITEMP2010 + mock_02_4000-4500_HITEMP2010:
            three .par files of ONE database, 5 isotopes, 3996.8-4003.2 and 6622-6667 cm-1
      ch4   mock_06_hit12                                      2 isotopes,  6622-6667 cm-1
    = 3 files / databases / species, 11 isotopes, 5 328 lines; and `multi`: ONE file with the
    three databases (the `onefile` run below).
  * `pb.run()` on tests/inputs/atmosphere_uniform_test.atm (51 layers, 9 species):
      wn_*    constant wavenumber step, CO2 + CH4 window (H2O database selected but empty)
      res_*   `resolution = 30000` on the same window         (_extcoeff.c:320-326, linterp)
      wl_*    `wlstep = 4e-5 um` on the same window           (same code path, another grid)
      hitemp  constant step on the dense two-file CO2 band head (280 lines per cm-1)
      wide    1.0 - 1.51 um at wnstep 1.0: all three species in one run, isotopes of three
              molecules with lines, mostly empty windows
      onefile the wn window from the ONE-file TLI: pins how the reference numbers the isotopes
              of a multi-database file (see RUNS)
    transit and emission for the first three; the emission runs use `quadrature = 5`
    (Gauss-Legendre angles, pyrat/spectrum.py:41-49).
  * stored per run: the grid definition, width grids and sizes of the Voigt table, the isotope
    tables Line_By_Line derived, the atmosphere, lbl.ec, od.depth, od.ideep, spectrum (+ the
    number and checksums of the lines selected).  NOT stored: `own` (recomputed from the
    definition), the line arrays (the TLI file is the fixture), the Voigt profiles.
  * `res_*` / `wl_*` additionally carry `ec_ieee`: the same unmodified sources compiled WITHOUT
    -ffast-math (the reference's setup.py:20 sets it).  In the interpolating modes the fast-math
    build re-associates `wn_i - (wn_0 + dw*ilo)` and loses an ulp that 1/dw amplifies
    (DESIGN.md section 2); the strict build shows what the source computes.

Only data is written: numbers the reference computed and bytes its TLI writer produced.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import tarfile
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_e2e import reference_package, REF      # noqa: E402

CFG_TLI = '''
[pyrat]
runmode = tli
logfile = {work}/{name}.log
tlifile = {work}/{name}.tli
dblist = {dblist}
dbtype = {dbtype}
pflist = {pflist}
wl_low = 1.00 um
wl_high = 2.51 um
verb = 0
'''

# one TLI file per molecule (how the reference's own configurations combine species:
# tests/configs/tli_multiple_opacity_{{H2O,CO2,CH4}}.cfg + `tlifile = a b c`), the CO2 one from
# three .par files of one database; and ONE file holding all of them (see ONEFILE below)
TLIS = {
    'h2o': ['{ref}/tests/inputs/Mock_HITRAN_H2O_1.00-1.01um.par'],
    'co2': ['{work}/mock_02_hit12.par', '{work}/mock_02_3750-4000_HITEMP2010.par',
            '{work}/mock_02_4000-4500_HITEMP2010.par'],
    'ch4': ['{work}/mock_06_hit12.par'],
}
TLIS['multi'] = TLIS['h2o'] + TLIS['co2'] + TLIS['ch4']

CFG_SPEC = '''
[pyrat]
runmode = spectrum
logfile = {work}/{name}.log
rt_path = {rt}
atmfile = {ref}/tests/inputs/atmosphere_uniform_test.atm
tlifile = {tlifiles}
radmodel = hydro_m
wl_low = {wl_low} um
wl_high = {wl_high} um
{sampling}
voigt_extent = 60.0
voigt_cutoff = 8.0
nlor = 24
ndop = 12
rstar = 1.27 rsun
tstar = 5800.0
mplanet = 0.6 mjup
rplanet = 1.0 rjup
refpressure = 0.1 bar
maxdepth = {maxdepth}
ncpu = 1
verb = 0
{extra}
'''

WINDOW = dict(wl_low='1.4995', wl_high='1.5105')
THREE = '{work}/h2o.tli\n    {work}/co2.tli\n    {work}/ch4.tli'
RUNS = [
    # name, rt, window, sampling keys, extra keys, TLI files, maxdepth (small where the window is
    # optically thin, so that columns still stop at different layers)
    ('wn_transit', 'transit', WINDOW, 'wnstep = 0.2\nwnosamp = 120', '', THREE, 0.02),
    ('wn_emission', 'emission', WINDOW, 'wnstep = 0.2\nwnosamp = 120', 'quadrature = 5', THREE,
     3e-4),
    ('res_transit', 'transit', WINDOW, 'resolution = 30000.0\nwnstep = 0.2\nwnosamp = 120', '',
     THREE, 0.02),
    ('res_emission', 'emission', WINDOW, 'resolution = 30000.0\nwnstep = 0.2\nwnosamp = 120',
     'quadrature = 5', THREE, 3e-4),
    ('wl_transit', 'transit', WINDOW, 'wlstep = 4e-5 um\nwnstep = 0.2\nwnosamp = 120', '', THREE,
     0.02),
    ('wl_emission', 'emission', WINDOW, 'wlstep = 4e-5 um\nwnstep = 0.2\nwnosamp = 120',
     'quadrature = 5', THREE, 3e-4),
    ('hitemp_transit', 'transit', dict(wl_low='2.4975', wl_high='2.5025'),
     'wnstep = 0.02\nwnosamp = 24', '', THREE, 0.15),
    ('wide_transit', 'transit', dict(wl_low='1.0', wl_high='1.51'),
     'wnstep = 1.0\nwnosamp = 720', '', THREE, 10.0),
    # ONE TLI file with the three databases.  The file stores every line's isotope index
    # relative to its own database (lread.py:181-209, 309) and Line_By_Line adds the isotope
    # count of the previous FILES only (line_by_line.py:114-119): the lines of the second and
    # third database of one file are computed with the isotope data of the first isotopes of
    # the file.  Pinned as the reference behaves (LBLSpectrum.from_tli(..., iso_numbering=
    # 'reference')); the default numbers isotopes over the file's databases.
    ('onefile_transit', 'transit', WINDOW, 'wnstep = 0.2\nwnosamp = 120', '', '{work}/multi.tli',
     10.0),
]


def strict_extcoeff(work):
    """The unmodified _extcoeff.c compiled without -ffast-math, as a module of its own."""
    import importlib.util
    import numpy
    out = os.path.join(work, 'strict')
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, '_extcoeff' + sysconfig.get_config_var('EXT_SUFFIX'))
    subprocess.check_call(
        ['gcc', '-shared', '-fPIC', '-O2', '-w', '-I' + sysconfig.get_paths()['include'],
         '-I' + numpy.get_include(), '-I' + os.path.join(REF, 'src_c', 'include'),
         os.path.join(REF, 'src_c', '_extcoeff.c'), '-o', so, '-lm'])
    spec = importlib.util.spec_from_file_location('_extcoeff', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def strict_ec(mod, pyrat, lbl):
    """lbl.ec again through the strict build, with the arguments of pyrat/extinction.py:197-208."""
    spec, atm, voigt = pyrat.spec, pyrat.atm, pyrat.voigt
    interpolate = spec.resolution is not None or spec.wlstep is not None
    ec = np.zeros_like(lbl.ec)
    for layer in range(atm.nlayers):
        row = np.zeros((1, spec.nwave))
        mod.extinction(row, voigt.profile, voigt.size, voigt.index, voigt.lorentz,
                       voigt.doppler, spec.wn, spec.own, spec.odivisors, atm.d[layer],
                       atm.mol_radius, atm.mol_mass, lbl.iso_atm_index, lbl.iso_mass,
                       lbl.iso_ratio, lbl.iso_pf[:, layer], np.copy(lbl.iso_mol_index), lbl.wn,
                       lbl.elow, lbl.gf, lbl.isoid, voigt.cutoff, lbl.ethresh,
                       atm.temp[layer], 0, 1, int(interpolate))
        ec[layer] = row[0]
    return ec


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_multi_')
    try:
        pb = reference_package(work)
        for name in ('mock_02_hit12', 'mock_06_hit12', 'mock_02_3750-4000_HITEMP2010',
                     'mock_02_4000-4500_HITEMP2010'):
            with tarfile.open(os.path.join(REF, 'tests', 'inputs', name + '.tar.gz')) as tar:
                tar.extract(name + '.par', work)
        strict = strict_extcoeff(work)

        def run(text, name, **kw):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(text.format(work=work, ref=REF, name=name, **kw))
            return pb.run(cfg)

        for tname, files in TLIS.items():
            run(CFG_TLI, tname, dblist='\n    '.join(files).format(work=work, ref=REF),
                dbtype=' '.join(['hitran'] * len(files)), pflist=' '.join(['tips'] * len(files)))
        # the H2O file is the one G13 already holds (same writer, same input): not stored twice
        with open(os.path.join(work, 'h2o.tli'), 'rb') as a, \
                open(os.path.join(HERE, 'g13_mock_h2o.tli'), 'rb') as b:
            wa, wb = a.read(), b.read()
        assert wa[:7] == wb[:7] and wa[23:] == wb[23:], 'H2O TLI differs from g13_mock_h2o.tli'
        for tname in ('co2', 'ch4', 'multi'):
            shutil.copy(os.path.join(work, tname + '.tli'), os.path.join(HERE, f'g16_{tname}.tli'))
        store = {}
        for name, rt, window, sampling, extra, tlifiles, maxdepth in RUNS:
            pyrat = run(CFG_SPEC, name, rt=rt, sampling=sampling, extra=extra,
                        tlifiles=tlifiles.format(work=work), maxdepth=maxdepth, **window)
            spec, atm, od, voigt = pyrat.spec, pyrat.atm, pyrat.od, pyrat.voigt
            lbl = pyrat.opacity.models[pyrat.opacity.models_type.index('lbl')]
            interp = spec.resolution is not None or spec.wlstep is not None
            mode = ('resolution' if spec.resolution is not None and name.startswith('res') else
                    'wlstep' if name.startswith('wl') else 'wnstep')
            r = dict(
                rt_path=rt, mode=mode, wn=spec.wn, wnlow=spec.wnlow, wnhigh=spec.wnhigh,
                wl_low=spec.wl_low, wl_high=spec.wl_high, wnstep=spec.wnstep,
                wnosamp=spec.wnosamp, ownstep=spec.ownstep, onwave=spec.onwave,
                own_first=spec.own[0], own_last=spec.own[-1], divisors=spec.odivisors,
                resolution=np.nan if spec.resolution is None else spec.resolution,
                wlstep=np.nan if spec.wlstep is None else spec.wlstep,
                lorentz=voigt.lorentz, doppler=voigt.doppler, size_out=voigt.size,
                index_out=voigt.index, extent=voigt.extent, cutoff=voigt.cutoff,
                dlratio=voigt.dlratio, nprofile=len(voigt.profile),
                profile_sum=np.sum(voigt.profile),
                ntli=len(lbl.tlifile), nlines=len(lbl.wn), lwn_sum=np.sum(lbl.wn), gf_sum=np.sum(lbl.gf),
                elow_sum=np.sum(lbl.elow), isoid_sum=np.sum(lbl.isoid.astype(np.int64)),
                iso_atm_index=lbl.iso_atm_index, iso_mass=lbl.iso_mass,
                iso_ratio=lbl.iso_ratio, iso_pf=lbl.iso_pf, iso_mol_index=lbl.iso_mol_index,
                lbl_species=np.array(list(lbl.species)), ethresh=lbl.ethresh,
                press=atm.press, temp=atm.temp, dens=atm.d, radius=atm.radius,
                mol_radius=atm.mol_radius, mol_mass=atm.mol_mass,
                species=np.array(list(atm.species)), rtop=atm.rtop, rstar=atm.rstar,
                maxdepth=od.maxdepth, ec=lbl.ec, depth=od.depth, ideep=od.ideep,
                spectrum=spec.spectrum)
            if rt == 'emission':
                r.update(quadrature=spec.quadrature, quadrature_mu=spec.quadrature_mu,
                         quadrature_weights=np.ravel(spec.quadrature_weights))
            if interp:
                r['ec_ieee'] = strict_ec(strict, pyrat, lbl)
                nz = lbl.ec != 0
                print('   fast-math vs strict build of the same source: max rel',
                      float(np.max(np.abs(r['ec_ieee'][nz] / lbl.ec[nz] - 1))))
            for k, v in r.items():
                store[f'{name}/{k}'] = v
            print(name, 'mode', mode, 'W', spec.nwave, 'onwave', spec.onwave, 'L', atm.nlayers,
                  'lines', len(lbl.wn), 'isotopes', len(lbl.iso_mass), 'species',
                  list(lbl.species), 'ideep', od.ideep.min(), od.ideep.max(), 'spectrum',
                  float(spec.spectrum.min()), float(spec.spectrum.max()))
        store['runs'] = np.array([r[0] for r in RUNS])
        np.savez_compressed(os.path.join(HERE, 'g16_multi.npz'), **store)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for f in sorted(os.listdir(HERE)):
        if f.startswith('g16_'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
