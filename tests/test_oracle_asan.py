"""The oracle's C restatement under AddressSanitizer + UBSan (CPU only): the extinction,
Voigt, column and continuum routines run over the golden inputs in a child process with
libasan preloaded; any out-of-bounds access or undefined behaviour aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import ctypes as C
import numpy as np
import oracle.oracle as o
o._LIB = C.CDLL(%(lib)r)            # the sanitizer build instead of liboracle.so
o._LIB.orc_trapezoid.restype = C.c_double
o._LIB.orc_simps.restype = C.c_double
o._LIB.orc_voigt_point.restype = C.c_double
o._LIB.orc_voigt_point.argtypes = [C.c_double] * 3
from tests import cases
from oracle import continuum as cont
# Voigt grid + every extinction variant of the golden case, both output-grid modes
for res in (False, True):
    c = cases.extinction_inputs(resolution=res)
    size = c['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    o.voigt_grid(profile, size, index, c['lorentz'], c['doppler'], c['own'][1] - c['own'][0])
    atm, iso = c['atm'], c['iso']
    for layer, add, cut, eth, skip in cases.extinction_variants():
        isoiext = iso['isoiext'].copy()
        if skip:
            isoiext[1] = -1
        temp = atm['temp'][layer]
        ext = np.zeros((1 if add else c['nspec'], len(c['wn'])))
        o.extinction(ext, profile, size, index, c['lorentz'], c['doppler'], c['wn'], c['own'],
                     c['divisors'], atm['dens'][layer], atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], cases.iso_z(temp, 3),
                     isoiext, c['lwn'], c['elow'], c['gf'], c['lid'],
                     c['cutoff'] if cut else 0.0, eth, temp, 0, add, int(res))
# the QUICK / Simpson / two-point-mean Voigt regimes
v = cases.voigt_case()
size = v['size'].copy()
index = np.zeros_like(size)
profile = np.zeros(np.sum(2 * size + 1))
o.voigt_grid(profile, size, index, v['lorentz'], v['doppler'], v['dwn'])
# table interpolation
t = cases.table_case()
nmol, ntemp, nlayers, nwave = t['etable'].shape
a = np.zeros((nlayers, nwave))
o.interp_ec(a, t['etable'], t['ttable'], t['temps'], t['dens'], 0, nlayers)
# columns
rng = np.random.default_rng(0)
ec = rng.uniform(1e-9, 1e-6, (12, 37))
radius = np.linspace(8e9, 7e9, 12)
depth, ideep = o.optical_depth_transit(ec, radius, 1, 11, 3.0)
o.transmission(depth, radius, 9e10, ideep, 1)
d2 = np.zeros((12, 37)); id2 = np.full(37, 11, np.int32)
o.plane_parallel_optical_depth(d2, id2, ec, -o.ediff(radius), 0.5, 0, 12)
wn = np.linspace(1000, 2000, 37)
temp = np.linspace(900, 1500, 12)
B = o.blackbody_wn_2D(wn, temp)
o.intensity(d2, id2, B, np.array([1.0, 0.5]), 0)
o.two_stream(np.cumsum(ec * 1e8, axis=0), wn, temp, np.ones(37), None, 0)
h = -o.ediff(radius)
hs, hr, hf = o.geth(h)
o.simps2D(B, h, np.full(37, 12, np.int32), hs, hr, hf)
# continuum pieces
g7 = np.load(%(root)r + '/tests/golden/g7_continuum.npz')
cont.cia_cross_section(g7['cia_h2he_tab'], g7['cia_h2he_temps'], g7['temp'], 0, 700)
y, x = g7['cia_raw_absorption'][1], g7['cia_raw_wn']
cont.splinterp_1D(y, x, cont.second_deriv(y, x), g7['wn'], 0.0)
det, mass, lpar, Z, cutoff = g7['alk_na_scalars']
cont.alkali_cross_section(g7['pressure'] * 1e6, g7['wn'], g7['temp'], g7['alk_na_voigt_det'], det,
                          mass, lpar, Z, cutoff, g7['alk_na_wn0'], g7['alk_na_gf'])
print('ASAN_CHILD_OK')
'''


def test_oracle_under_asan(tmp_path):
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True,
                             text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip('libasan not available')
    build = subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle'), 'liboracle_asan.so'],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    lib = os.path.join(ROOT, 'oracle', 'liboracle_asan.so')
    env = dict(os.environ, LD_PRELOAD=libasan,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=1:halt_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    out = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'lib': lib}],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert out.returncode == 0 and 'ASAN_CHILD_OK' in out.stdout, (out.stdout[-1500:]
                                                                    + out.stderr[-3000:])
