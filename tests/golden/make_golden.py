#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference C extensions.

Run in the build container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Each fixture stores the inputs and the outputs of the compiled reference
(pyratbay v2.0.1 src_c, flags -O3 -ffast-math as in the reference's setup.py:20).
Only data is written -- no reference source text.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import cases                                  # noqa: E402
from oracle import ref                        # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def g1_voigt():
    vp = ref.module('vprofile')
    c = cases.voigt_case()
    size = c['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    vp.grid(profile, size, index, c['lorentz'], c['doppler'], c['dwn'], 0)
    used = int(np.sum(2 * c['size'][c['size'] > 0] + 1))
    # the >99999-sample QUICK cell is stored as a strided subsample + checksum
    q0 = int(index[5, 0])
    qn = 2 * int(size[5, 0]) + 1
    np.savez_compressed(
        os.path.join(OUT, 'g1_voigt.npz'),
        lorentz=c['lorentz'], doppler=c['doppler'], dwn=c['dwn'], size_in=c['size'],
        size_out=size, index_out=index, used=used,
        profile_head=profile[:q0], quick_start=q0, quick_n=qn,
        quick_stride=41, quick_sub=profile[q0:q0 + qn:41],
        quick_sum=np.sum(profile[q0:q0 + qn]))


def _voigt_table(c, vp):
    size = c['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    ownstep = c['own'][1] - c['own'][0]
    vp.grid(profile, size, index, c['lorentz'], c['doppler'], ownstep, 0)
    return profile, size, index


def g2_extinction():
    ec = ref.module('_extcoeff')
    vp = ref.module('vprofile')
    store = {}
    for mode, resolution in (('step', 0), ('res', 1)):
        c = cases.extinction_inputs(resolution=bool(resolution))
        profile, size, index = _voigt_table(c, vp)
        atm, iso = c['atm'], c['iso']
        outs = []
        for (layer, add, cut, eth, skip) in cases.extinction_variants():
            isoiext = iso['isoiext'].astype(int).copy()
            if skip:
                isoiext[1] = -1
            temp = atm['temp'][layer]
            z = cases.iso_z(temp, len(isoiext))
            ext = np.zeros((1 if add else c['nspec'], len(c['wn'])))
            ec.extinction(
                ext, profile, size, index, c['lorentz'], c['doppler'],
                c['wn'], c['own'], c['divisors'].astype(int),
                atm['dens'][layer], atm['mol_radius'], atm['mol_mass'],
                iso['isoimol'].astype(int), iso['isomass'], iso['isoratio'], z, isoiext,
                c['lwn'], c['elow'], c['gf'], c['lid'].astype(int),
                c['cutoff'] if cut else 0.0, eth, temp, 0, add, resolution)
            full = np.zeros((c['nspec'], len(c['wn'])))
            full[:ext.shape[0]] = ext
            outs.append(full)
        store[f'ext_{mode}'] = np.array(outs)
        store[f'size_out_{mode}'] = size
        store[f'index_out_{mode}'] = index
    np.savez_compressed(os.path.join(OUT, 'g2_extinction.npz'),
                        variants=np.array(cases.extinction_variants(), float), **store)


def g3_interp():
    ec = ref.module('_extcoeff')
    c = cases.table_case()
    nmol, ntemp, nlayers, nwave = c['etable'].shape
    a = np.full((nlayers, nwave), 1e-12)
    ec.interp_ec(a, c['etable'], c['ttable'], c['temps'], c['dens'], 0, nlayers)
    b = np.zeros((nlayers, nwave))
    ec.interp_ec(b, c['etable'], c['ttable'], c['temps'], c['dens'], 2, 5)
    m = np.zeros((nmol, nlayers, nwave))
    ec.interp_ec_per_mol(m, c['etable'], c['ttable'], c['temps'], c['dens'], 0, nlayers + 3)
    np.savez_compressed(os.path.join(OUT, 'g3_interp.npz'), full=a, part=b, per_mol=m,
                        **c)


def g4_depth():
    t = ref.module('_trapezoid')
    cu = ref.module('cutils')
    from oracle.oracle import transit_path
    c = cases.column_case()
    L, W = c['nlayers'], c['nwave']
    store = dict(radius=c['radius'], ec=c['ec'])
    for tag, itop, ibottom, maxdepth in (('a', 0, L, 10.0), ('b', 3, L - 2, 10.0),
                                         ('c', 2, L, np.inf)):
        raypath = transit_path(c['radius'], itop)
        depth = np.zeros((L, W))
        ideep = np.full(W, -1, np.intc)
        r = itop
        for r in range(itop, ibottom):
            depth[r] = t.optdepth(c['ec'][itop:r + 1], raypath[r], maxdepth, ideep, r)
        ideep[ideep < 0] = r
        store[f'transit_depth_{tag}'] = depth
        store[f'transit_ideep_{tag}'] = ideep
        h = -cu.ediff(c['radius'])
        pdepth = np.zeros((L, W))
        pideep = np.full(W, L - 1)
        t.plane_parallel_optical_depth(pdepth, pideep, c['ec'], h, maxdepth, itop, ibottom)
        store[f'plane_depth_{tag}'] = pdepth
        store[f'plane_ideep_{tag}'] = pideep
        store[f'args_{tag}'] = np.array([itop, ibottom, maxdepth])
    np.savez_compressed(os.path.join(OUT, 'g4_depth.npz'), **store)


def g5_rt():
    t = ref.module('_trapezoid')
    bb = ref.module('_blackbody')
    sm = ref.module('_simpson')
    cu = ref.module('cutils')
    ind = ref.module('_indices')
    c = cases.column_case()
    L, W = c['nlayers'], c['nwave']
    g4 = np.load(os.path.join(OUT, 'g4_depth.npz'))
    store = dict(wn=c['wn'], temp=c['temp'], mu=c['mu'], radius=c['radius'],
                 rstar=c['rstar'])
    # Planck
    store['B_full'] = bb.blackbody_wn_2D(c['wn'], c['temp'])
    last = (np.arange(W) % L).astype(np.intc)
    B = np.zeros((L, W))
    bb.blackbody_wn_2D(c['wn'], c['temp'], B, last)
    store['B_last'] = B
    store['last'] = last
    store['B_1d'] = bb.blackbody_wn(c['wn'], 1234.5)
    # emission intensity for the three plane-parallel depth fields
    for tag in 'abc':
        itop = int(g4[f'args_{tag}'][0])
        ideep = g4[f'plane_ideep_{tag}'].astype(np.intc).copy()
        if tag == 'b':
            ideep[:6] = itop + 1          # exercises the last-rtop == 1 branch
        store[f'intensity_{tag}'] = t.intensity(
            g4[f'plane_depth_{tag}'], ideep, store['B_full'], c['mu'], itop)
        store[f'intensity_ideep_{tag}'] = ideep
        # transmission integral (radiative_transfer.py:57-71)
        tdepth = g4[f'transit_depth_{tag}']
        tideep = g4[f'transit_ideep_{tag}']
        h = np.ediff1d(c['radius'][itop:])
        integ = np.exp(-tdepth[itop:]) * np.expand_dims(c['radius'][itop:], 1)
        nint = (tideep - itop).astype(np.intc)
        trap = t.trapezoid2D(integ, h, nint)
        store[f'trap2d_{tag}'] = trap
        store[f'transmission_{tag}'] = (c['radius'][itop]**2 + 2 * trap) / c['rstar']**2
    # 1D helpers
    y = np.sin(np.linspace(0.1, 2.0, 12))**2 + 0.3
    h = np.diff(np.linspace(0.1, 2.0, 12)**1.3)
    store['trap1d_y'] = y
    store['trap1d_h'] = h
    store['trap1d'] = t.trapezoid(y, h)
    out = np.zeros(12)
    store['cumsum_n'] = t.cumulative_sum(out, y, h, 0.9)
    store['cumsum_out'] = out
    store['ediff'] = cu.ediff(c['radius'])
    # Simpson, odd and even sample counts
    rng = np.random.default_rng(5)
    for tag, ny in (('odd', 11), ('even', 12)):
        x = np.sort(rng.uniform(0, 3, ny))
        hh = np.diff(x)
        yy = np.exp(-x)[:, None] * rng.uniform(0.5, 2.0, (ny, 17))
        hs, hr, hf = sm.geth(hh)
        store[f'simps_x_{tag}'] = x
        store[f'simps_y_{tag}'] = yy
        store[f'simps_hsum_{tag}'] = hs
        store[f'simps_hratio_{tag}'] = hr
        store[f'simps_hfactor_{tag}'] = hf
        store[f'simps_1d_{tag}'] = sm.simps(yy[:, 0].copy(), hh, hs, hr, hf)
        nint = rng.integers(0, ny + 1, 17).astype(np.intc)
        nint[0] = ny
        nint[1] = 2
        nint[2] = 1
        # geth pairs depend on the parity of len(h); simps2D uses one set for all columns
        store[f'simps_nint_{tag}'] = nint
        store[f'simps_2d_{tag}'] = sm.simps2D(yy, hh, nint, hs, hr, hf)
    flags = np.array([0, 0, 1, 0, 1, 1, 0], np.intc)
    store['ifirst'] = np.array([ind.ifirst(flags), ind.ifirst(np.zeros(4, np.intc), -7)])
    store['ilast'] = np.array([ind.ilast(flags), ind.ilast(np.zeros(4, np.intc))])
    store['flags'] = flags
    np.savez_compressed(os.path.join(OUT, 'g5_rt.npz'), **store)


if __name__ == '__main__':
    if not ref.available():
        sys.exit('oracle/_ref is not built: run `make -C oracle ref` first')
    g1_voigt()
    g2_extinction()
    g3_interp()
    g4_depth()
    g5_rt()
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')
