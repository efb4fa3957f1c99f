"""Randomised sweep of the `resolution` mode (GPU box): the per-layer dynamic grids (gather mode
'dynamic') against the direct gather of the same plan (1e-12 of every sample, equal zero pattern)
and against the compiled reference (1e-10) -- random grids (resolving power, fine-grid factor),
atmospheres (pressure range, shuffled layers: factors that come back), line lists, add 0/1,
several rows, skipped isotopes, ethresh up to 1e-2, cutoff on/off, wavenumber shards (bit-equal to
the slice of the whole call), a changed atmosphere on the same plan (new factors, new Lorentz rows).
usage: python tools/fuzz_resdyn.py [count] [seed0]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def host(t):
    return t.cpu().numpy()


relaxed = [0]


def one(eng, ref, rng, seed):
    import torch
    from oracle import oracle as orc
    from pyratbay_amd import synth
    nwave = int(rng.integers(200, 5000))
    nlayers = int(rng.integers(1, 14))
    nlines = int(rng.integers(50, 10000))
    niso = int(rng.integers(1, 4))
    osamp = int(rng.choice([12, 24, 60, 120]))
    res = float(rng.choice([2.0e4, 6.0e4, 1.5e5]))
    case = synth.lbl_case(nwave, nlayers, nlines, wnstep=float(rng.choice([0.02, 0.05])),
                          wnosamp=osamp, nlor=int(rng.integers(6, 20)), ndop=int(rng.integers(3, 9)),
                          extent=float(rng.choice([8.0, 40.0, 150.0])),
                          cutoff=float(rng.choice([0.0, 0.5, 3.0, 30.0])), niso=niso, seed=seed,
                          ptop=10.0**rng.uniform(-7, -3), pbottom=10.0**rng.uniform(-1, 2),
                          resolution=res)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    add = bool(rng.random() < 0.5)
    ethresh = float(rng.choice([1e-30, 1e-6, 1e-2]))
    isoiext = np.array(iso['isoiext'], np.int32).copy()
    if not add:
        isoiext = rng.integers(0, int(rng.integers(1, 4)), niso).astype(np.int32)
        isoiext[rng.integers(0, niso)] = 0
    if niso > 1 and rng.random() < 0.3:
        isoiext[rng.integers(1, niso)] = -1
    rows = 1 if add else int(isoiext.max()) + 1
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                              g['wnosamp'], 2)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], isoiext, vg['cutoff'], ethresh,
                  resolution=True, max_layers=nlayers)
    order = rng.permutation(nlayers) if rng.random() < 0.5 else np.arange(nlayers)
    temp, dens, isoz = atm['temp'][order], atm['dens'][order], iso['isoz'][:, order]
    factors = set()
    for trial in range(2):
        t, d, z = eng.dev(temp), eng.dev(dens), eng.dev(isoz)
        lbl.set_gather_mode('auto')
        want = lbl.extinction(t, d, z, add=add)
        lbl.set_gather_mode('dynamic')
        got = lbl.extinction(t, d, z, add=add)
        assert lbl.last_gather_kernel == ('dynamic grids' if ll.ngroups > 0 else 'k_ext_linterp')
        factors |= set(lbl.last_state(nlayers, rows)[0].tolist())
        w, h = host(want), host(got)
        assert np.array_equal(w == 0, h == 0), 'dynamic vs direct: zero pattern'
        np.testing.assert_allclose(h, w, rtol=1e-12, err_msg='dynamic vs direct')
        if g['nwave'] > 300:
            a = int(rng.integers(0, g['nwave'] - 100))
            b = int(rng.integers(a + 1, g['nwave'] + 1))
            part = lbl.extinction(t, d, z, add=add, wbegin=a, wcount=b - a)
            assert torch.equal(part, got[:, :, a:b]), 'shard != slice of the whole call'
        if trial == 0:
            profile, psize, pindex = vt.flat(), np.array(vt.size), np.array(vt.index)
            E = ref.module('_extcoeff')
            for k in range(nlayers):
                r = np.zeros((rows, g['nwave']))
                E.extinction(r, profile, psize, pindex, vg['lorentz'], vg['doppler'], g['wn'],
                             g['own'], g['divisors'], dens[k], atm['mol_radius'],
                             atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                             isoz[:, k].copy(), isoiext, ln['lwn'], ln['elow'], ln['gf'],
                             ln['lid'], vg['cutoff'], ethresh, float(temp[k]), 0, int(add), 1)
                assert np.array_equal(h[k] == 0, r == 0), 'vs reference: zero pattern'
                nz = r != 0
                if nz.any() and np.max(np.abs(h[k][nz] / r[nz] - 1)) > 1e-10:
                    # the reference's -ffast-math build re-associates the interpolation weights
                    # (DESIGN.md section 2): the same expressions in strict IEEE order decide
                    strict = np.zeros_like(r)
                    orc.extinction(strict, profile, psize, pindex, vg['lorentz'], vg['doppler'],
                                   g['wn'], g['own'], g['divisors'], dens[k], atm['mol_radius'],
                                   atm['mol_mass'], iso['isoimol'], iso['isomass'],
                                   iso['isoratio'], isoz[:, k].copy(), isoiext, ln['lwn'],
                                   ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], ethresh,
                                   float(temp[k]), 0, int(add), 1)
                    np.testing.assert_allclose(h[k], strict, rtol=1e-10, err_msg='vs strict oracle')
                    np.testing.assert_allclose(r, strict, rtol=1e-6, err_msg='reference vs oracle')
                    relaxed[0] += 1
            # another atmosphere on the same plan: hotter / denser, other factors and table rows
            temp = temp * rng.uniform(0.6, 1.6)
            dens = dens * 10.0**rng.uniform(-1.0, 1.0)
    lbl.close()
    ll.close()
    vt.close()
    return len(factors)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 52000
    from oracle import ref
    if not ref.available():
        sys.exit('oracle/_ref is not built')
    from pyratbay_amd import engine
    engine.require_gpu()
    bad, nf = [], []
    for i in range(count):
        seed = seed0 + i
        try:
            nf.append(one(engine, ref, np.random.default_rng(seed), seed))
        except Exception:                                  # noqa: BLE001
            bad.append(seed)
            print('FAIL seed', seed)
            traceback.print_exc(limit=4)
        if i % 50 == 49:
            print(f'{i + 1} seeds, {len(bad)} failures', flush=True)
    print(f'{count} seeds, {np.mean(nf) if nf else 0:.1f} factors per plan on average '
          f'(largest {max(nf) if nf else 0}); {relaxed[0]} layers judged by the strict-IEEE oracle '
          f'(the fast-math reference build is off by more than 1e-10 there)')
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
