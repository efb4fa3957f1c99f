"""One-off sweep of the compute_opacity batch (opacity_table.py: every (T, p) cell of a sampled
cross-section table in chunked launches) against the compiled reference's per-layer extinction
with add = 0, and of the .npz round trip.  usage: python tools/fuzz_opacity.py [count]"""
import os
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(eng, ref, rng, seed, tmp):
    from pyratbay_amd import synth, opacity_table as ot
    niso = int(rng.integers(1, 4))
    osamp = int(rng.choice([6, 12, 24]))
    nl = int(rng.integers(1, 8))
    case = synth.lbl_case(int(rng.integers(3, 2500)), nl, int(rng.integers(1, 6000)),
                          wnstep=float(rng.choice([0.05, 0.2])), wnosamp=osamp, nlor=12, ndop=6,
                          extent=float(rng.choice([8.0, 40.0, 150.0])),
                          cutoff=float(rng.choice([0.5, 3.0, 30.0])), niso=niso, seed=seed)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], osamp, True)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], niso, g['own'])
    lbl = eng.LBL(vt, ll, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                  iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'],
                  1e-30, max_layers=int(rng.integers(1, 40)))
    ntemp = int(rng.integers(1, 6))
    tgrid = np.sort(rng.uniform(300.0, 3000.0, ntemp))
    pf = np.array([synth.partition_function(tgrid)] * niso)
    nw = g['nwave']
    chunk = int(rng.choice([nw * 8, 5 * nw * 8, 1 << 30]))
    etable = ot.compute_opacity(lbl, tgrid, atm['press'], atm['vmr'], pf, chunk_bytes=chunk)
    got = etable.cpu().numpy()
    assert got.shape == (1, ntemp, nl, nw)
    profile, psize, pindex = vt.flat(), np.array(vt.size), np.array(vt.index)
    E = ref.module('_extcoeff')
    cells = [(int(rng.integers(0, ntemp)), int(rng.integers(0, nl))) for _ in range(4)]
    for it, il in set(cells):
        t = float(tgrid[it])
        dens = atm['vmr'][il] * atm['press'][il] * synth.BAR / (synth.K_B * t)
        want = np.zeros((1, nw))
        E.extinction(want, profile, psize, pindex, vg['lorentz'], vg['doppler'], g['wn'],
                     g['own'], g['divisors'], dens, atm['mol_radius'], atm['mol_mass'],
                     iso['isoimol'], iso['isomass'], iso['isoratio'], pf[:, it].copy(),
                     iso['isoiext'], ln['lwn'], ln['elow'], ln['gf'], ln['lid'], vg['cutoff'],
                     1e-30, t, 0, 0, 0)
        assert np.array_equal(got[0, it, il] == 0, want[0] == 0), 'zero pattern'
        np.testing.assert_allclose(got[0, it, il], want[0], rtol=1e-10)
    path = os.path.join(tmp, f't{seed}.npz')
    ot.write_opacity(path, 'H2O', tgrid, atm['press'], g['wn'], etable[0])
    units, species, t_, p_, w_, o_ = ot.read_opacity(path)
    assert species == 'H2O' and np.array_equal(o_, got[0]) and np.array_equal(t_, tgrid)
    os.remove(path)
    lbl.close()
    ll.close()
    vt.close()


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    from oracle import ref
    from pyratbay_amd import engine
    engine.require_gpu()
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(count):
            try:
                one(engine, ref, np.random.default_rng(110000 + seed), seed, tmp)
            except Exception:                              # noqa: BLE001
                bad.append(seed)
                print('FAIL seed', seed)
                traceback.print_exc(limit=3)
            if seed % 50 == 49:
                print(f'{seed + 1} seeds, {len(bad)} failures', flush=True)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
