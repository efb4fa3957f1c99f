#!/usr/bin/env python3
"""Generate tests/golden/g6_e2e_*.npz by running the REAL reference package end to end.

Build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden_e2e.py

What it does, all inside a temporary directory that is deleted afterwards:
  * copies /root/reference/pyratbay and drops the compiled, unmodified C extensions
    (oracle/_ref/*.so) into pyratbay/lib/;
  * provides import stand-ins for the three third-party packages that are not installed
    here and are not on this path (mc3: logging/plot themes only; chemcat; h5py) --
    these stubs are part of THIS script, they contain no reference code;
  * `pb.run()` a TLI build from the reference's bundled mock HITRAN file
    (tests/inputs/Mock_HITRAN_H2O_1.00-1.01um.par, 888 H2O lines, 4 isotopes) and then a
    line-by-line transmission and an emission spectrum on the reference's own test
    atmosphere (tests/inputs/atmosphere_uniform_test.atm);
  * stores the arrays the package hands to its C extensions (inputs) and what it gets
    back / derives (lbl.ec, od.depth, od.ideep, spectrum) as the fixture.

Only data is written: no reference source text.
"""
import os
import shutil
import sys
import tempfile
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))

STUBS = {
    'mc3/__init__.py': '''
        from . import utils, plots, stats
        def sample(*a, **k):
            raise NotImplementedError('mc3 is not installed (stub)')
    ''',
    'mc3/utils.py': '''
        import sys
        class Log:
            def __init__(self, logname=None, verb=2, append=False, width=70):
                self.logname, self.verb, self.width = logname, verb, width
                self.file, self.sep = None, 70*':'
                self.warnings = []
            def msg(self, text, verb=2, indent=0, **k): pass
            def head(self, text, indent=0, **k): pass
            def debug(self, text, indent=0, **k): pass
            def warning(self, text, **k): self.warnings.append(text)
            def error(self, text, exception=ValueError, **k): raise exception(text)
            def close(self): pass
        def burn(*a, **k):
            raise NotImplementedError
    ''',
    'mc3/plots.py': '''
        class Theme:
            def __init__(self, color='blue', *a, **k): self.color = color
        THEMES = {name: Theme(name) for name in
                  ('blue', 'red', 'black', 'green', 'orange', 'indigo', 'purple')}
        class Posterior:
            def __init__(self, *a, **k): raise NotImplementedError
        def trace(*a, **k): raise NotImplementedError
    ''',
    'mc3/stats.py': '''
        def __getattr__(name):
            def missing(*a, **k):
                raise NotImplementedError(f'mc3.stats.{name} (stub)')
            return missing
    ''',
    'chemcat/__init__.py': '''
        class Network:
            def __init__(self, *a, **k): raise NotImplementedError('chemcat stub')
    ''',
    'h5py/__init__.py': '''
        class File:
            def __init__(self, *a, **k): raise NotImplementedError('h5py stub')
    ''',
}

CFG_TLI = '''
[pyrat]
runmode = tli
logfile = {work}/mock_h2o.log
tlifile = {work}/mock_h2o.tli
dblist = {ref}/tests/inputs/Mock_HITRAN_H2O_1.00-1.01um.par
dbtype = hitran
pflist = tips
wl_low = 1.00 um
wl_high = 1.01 um
verb = 0
'''

CFG_SPEC = '''
[pyrat]
runmode = spectrum
logfile = {work}/spec_{rt}.log
rt_path = {rt}
atmfile = {ref}/tests/inputs/atmosphere_uniform_test.atm
tlifile = {work}/mock_h2o.tli
radmodel = hydro_m
wl_low = 1.00 um
wl_high = 1.01 um
wnstep = 0.1
wnosamp = 60
voigt_extent = 60.0
voigt_cutoff = 8.0
nlor = 24
ndop = 12
rstar = 1.27 rsun
tstar = 5800.0
mplanet = 0.6 mjup
rplanet = 1.0 rjup
refpressure = 0.1 bar
maxdepth = 10.0
ncpu = 1
verb = 0
'''


def reference_package(work):
    """Import the real package from a scratch copy under `work` (unmodified C extensions
    compiled in place, import stand-ins for mc3/chemcat/h5py) and return the module."""
    shutil.copytree(os.path.join(REF, 'pyratbay'), os.path.join(work, 'pkg', 'pyratbay'))
    os.system(f'chmod -R u+w {work}')
    # unmodified reference extensions (all ten .c files of src_c)
    import subprocess
    import sysconfig
    import numpy
    ext = sysconfig.get_config_var('EXT_SUFFIX')
    libdir = os.path.join(work, 'pkg', 'pyratbay', 'lib')
    for name in ('_extcoeff', 'vprofile', '_trapezoid', '_simpson', '_blackbody', 'cutils',
                 '_indices', '_alkali', '_pt', '_spline'):
        subprocess.check_call(
            ['gcc', '-shared', '-fPIC', '-O3', '-ffast-math', '-w',
             '-I' + sysconfig.get_paths()['include'], '-I' + numpy.get_include(),
             '-I' + os.path.join(REF, 'src_c', 'include'),
             os.path.join(REF, 'src_c', name + '.c'), '-o',
             os.path.join(libdir, name + ext), '-lm'])
    for rel, text in STUBS.items():
        path = os.path.join(work, 'stubs', rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(textwrap.dedent(text))
    sys.path.insert(0, os.path.join(work, 'pkg'))
    sys.path.insert(0, os.path.join(work, 'stubs'))
    import matplotlib
    matplotlib.use('Agg')
    import pyratbay as pb
    return pb


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_e2e_')
    try:
        pb = reference_package(work)

        def run(cfg_text, name, **kw):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(cfg_text.format(work=work, ref=REF, **kw))
            return pb.run(cfg)

        run(CFG_TLI, 'tli')
        for rt in ('transit', 'emission', 'emission_two_stream'):
            extra = 'smaxis = 0.045 au\ntint = 150.0\nbeta_irr = 0.25\n' if 'two_stream' in rt else ''
            pyrat = run(CFG_SPEC + extra, f'spec_{rt}', rt=rt)
            spec, atm, od, voigt = pyrat.spec, pyrat.atm, pyrat.od, pyrat.voigt
            lbl = pyrat.opacity.models[pyrat.opacity.models_type.index('lbl')]
            store = dict(
                rt_path=rt,
                wn=spec.wn, own=spec.own, divisors=spec.odivisors, wnosamp=spec.wnosamp,
                ownstep=spec.ownstep, onwave=spec.onwave,
                lorentz=voigt.lorentz, doppler=voigt.doppler, size_out=voigt.size,
                index_out=voigt.index, extent=voigt.extent, cutoff=voigt.cutoff,
                dlratio=voigt.dlratio, nprofile=len(voigt.profile),
                profile_sub=voigt.profile[::37], profile_sum=np.sum(voigt.profile),
                lwn=lbl.wn, elow=lbl.elow, gf=lbl.gf, isoid=lbl.isoid,
                iso_atm_index=lbl.iso_atm_index, iso_mass=lbl.iso_mass,
                iso_ratio=lbl.iso_ratio, iso_pf=lbl.iso_pf, iso_mol_index=lbl.iso_mol_index,
                ethresh=lbl.ethresh,
                press=atm.press, temp=atm.temp, dens=atm.d, radius=atm.radius,
                mol_radius=atm.mol_radius, mol_mass=atm.mol_mass, rtop=atm.rtop,
                rstar=atm.rstar, maxdepth=od.maxdepth,
                ec=lbl.ec, depth=od.depth, ideep=od.ideep, spectrum=spec.spectrum,
            )
            if rt == 'emission':
                store.update(quadrature_mu=spec.quadrature_mu,
                             quadrature_weights=np.ravel(spec.quadrature_weights),
                             intensity=spec.intensity)
            if rt == 'emission_two_stream':
                # the two-stream solver's own inputs and outputs; ec stays out (same as
                # the emission fixture) to keep the file small
                del store['ec']
                for key in ('lwn', 'elow', 'gf', 'isoid', 'profile_sub', 'own'):
                    store.pop(key, None)
                store.update(tint=atm.tint, beta_irr=atm.beta_irr, smaxis=atm.smaxis,
                             starflux=spec.starflux, f_int=spec.f_int,
                             flux_up=spec.flux_up, flux_down=spec.flux_down,
                             exp1_x=np.logspace(-8, 2.4, 300),
                             exp1_y=__import__('scipy.special', fromlist=['exp1']).exp1(
                                 np.logspace(-8, 2.4, 300)))
            np.savez_compressed(os.path.join(OUT, f'g6_e2e_{rt}.npz'), **store)
            print(rt, 'W', spec.nwave, 'L', atm.nlayers, 'lines', len(lbl.wn),
                  'spectrum', spec.spectrum.min(), spec.spectrum.max())
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for f in sorted(os.listdir(OUT)):
        if f.startswith('g6_'):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
