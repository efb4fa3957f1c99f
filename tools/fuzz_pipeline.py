"""One-off sweep of the whole path on random cases: LBLSpectrum (transit / emission / two-stream)
against the oracle chain, wavenumber shards of random world sizes concatenated, the two-phase shard
form (records of the shard's groups + maxima exchanged), layer shards interleaved, the HIP-graph
replay and set_atmosphere.  usage: python tools/fuzz_pipeline.py [count]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RTOL = 1e-10


def host(t):
    return t.cpu().numpy()


def one(eng, orc, rng, seed):
    import torch

    def mark(what):
        if os.environ.get('PB_FUZZ_TRACE'):
            torch.cuda.synchronize()
            print(f'   seed {seed}: {what} ok', flush=True)

    from pyratbay_amd import synth
    from pyratbay_amd.dist import shard_bounds
    nwave = int(rng.integers(8, 5000))
    nl = int(rng.integers(2, 20))
    case = synth.lbl_case(nwave, nl, int(rng.integers(1, 20000)),
                          wnstep=float(rng.choice([0.01, 0.05, 0.2])),
                          wnosamp=int(rng.choice([6, 12, 24, 60])), nlor=12, ndop=6,
                          extent=float(rng.choice([8.0, 40.0, 150.0])),
                          cutoff=float(rng.choice([0.5, 3.0, 30.0])),
                          niso=int(rng.integers(1, 4)), seed=seed)
    case['ethresh'] = float(rng.choice([1e-30, 1e-4]))
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    nw = g['nwave']
    rt = str(rng.choice(['transit', 'emission']))
    single = eng.LBLSpectrum(case, rt_path=rt)
    spec = host(single.run())
    mark(f'single run {rt} nw={nw} nl={nl}')
    # oracle chain on every layer
    profile = single.voigt.flat()
    ec = np.zeros((nl, nw))
    for k in range(nl):
        row = np.zeros((1, nw))
        orc.extinction(row, profile, single.voigt.size, single.voigt.index, vg['lorentz'],
                       vg['doppler'], g['wn'], g['own'], g['divisors'], atm['dens'][k],
                       atm['mol_radius'], atm['mol_mass'], iso['isoimol'], iso['isomass'],
                       iso['isoratio'], iso['isoz'][:, k].copy(), iso['isoiext'], ln['lwn'],
                       ln['elow'], ln['gf'], ln['lid'], vg['cutoff'], case['ethresh'],
                       atm['temp'][k], 0, 1, 0)
        ec[k] = row[0]
    got_ec = host(single.ec)[:, 0]
    assert np.array_equal(got_ec == 0, ec == 0), 'ec zero pattern'
    np.testing.assert_allclose(got_ec, ec, rtol=RTOL)
    if rt == 'transit':
        depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, nl, case['maxdepth'])
        want = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    else:
        depth = np.zeros((nl, nw))
        ideep = np.full(nw, nl - 1, np.int32)
        orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(atm['radius']),
                                         case['maxdepth'], 0, nl)
        inten = orc.intensity(depth, ideep, orc.blackbody_wn_2D(g['wn'], atm['temp']),
                              host(single.mu), 0)
        want = np.sum(inten * host(single.weights)[:, None], axis=0)
    np.testing.assert_allclose(spec, want, rtol=RTOL)
    # wavenumber shards, one-call and two-phase form
    world = int(rng.integers(2, 9))
    b = shard_bounds(nw, world)
    maxima = []
    models = []
    for r in range(world):
        wc = int(b[r + 1] - b[r])
        if wc == 0:
            continue
        m = eng.LBLSpectrum(case, rt_path=rt, wbegin=int(b[r]), wcount=wc, voigt=single.voigt,
                            lines=single.lines)
        models.append((r, m))
    if os.environ.get('PB_FUZZ_TRACE') == '2':
        parts = []
        for r, m in models:
            mark(f'shard {r} [{m.wbegin}, {m.wbegin + m.wcount}) created')
            m.extinction()
            mark(f'shard {r} extinction ({m.lbl.last_gather_kernel})')
            m.run()
            mark(f'shard {r} run')
            parts.append(host(m.spectrum))
    else:
        parts = [host(m.run()) for _, m in models]
    mark('shard runs')
    assert np.array_equal(np.concatenate(parts), spec), 'wavenumber shards'
    # two-phase: every shard derives its own records, the maxima are combined by hand
    for _, m in models:
        m.lbl.extinction_begin(m.temp, m.dens, m.isoz, add=True, out=m.ec, wbegin=m.wbegin,
                               wcount=m.wcount)
        maxima.append(m.lbl.kmax_tensor().clone())
    mark('extinction_begin of every shard')
    top = torch.stack(maxima).max(dim=0).values
    parts2 = []
    for _, m in models:
        m.lbl.kmax_tensor().copy_(top)
        m.lbl.extinction_end()
        parts2.append(host(m.ec)[:, 0])
    mark('extinction_end of every shard')
    assert np.array_equal(np.concatenate(parts2, axis=1), got_ec), 'two-phase shards'
    # layer shards interleaved
    lw = int(rng.integers(2, 6))
    if not os.environ.get('PB_FUZZ_OLD'):
        lw = min(lw, nl)
    got = np.zeros_like(got_ec)
    for r in range(lw):
        idx = torch.arange(r, nl, lw, device='cuda')
        if len(idx):
            e = single.lbl.extinction(single.temp[idx].contiguous(), single.dens[idx].contiguous(),
                                      single.isoz[:, idx].contiguous(), add=True)
            got[r::lw] = host(e)[:, 0]
    mark('layer shards')
    np.testing.assert_allclose(got, got_ec, rtol=1e-12)      # the phase split may differ
    assert np.array_equal(got == 0, got_ec == 0)
    # graph replay after an in-place atmosphere update
    hot = atm['temp'] * 1.03
    zhot = synth.partition_function(hot)[None, :].repeat(len(iso['isomass']), 0)
    replay = single.capture()
    mark('capture')
    assert np.array_equal(host(replay()), spec), 'graph replay'
    single.set_atmosphere(hot, atm['dens'], zhot)
    r1 = host(replay()).copy()
    mark('replay after set_atmosphere')
    fresh = eng.LBLSpectrum(case, rt_path=rt, voigt=single.voigt, lines=single.lines)
    fresh.set_atmosphere(hot, atm['dens'], zhot)
    assert np.array_equal(r1, host(fresh.run())), 'graph replay after set_atmosphere'


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    from oracle import oracle
    oracle.lib()
    from pyratbay_amd import engine, _capi
    if os.environ.get('PB_PROBE_LIB'):          # an instrumented build (tools only)
        _capi.LIBPATH = os.path.abspath(os.environ['PB_PROBE_LIB'])
    engine.require_gpu()
    bad = []
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for seed in range(first, first + count):
        try:
            one(engine, oracle, np.random.default_rng(12000 + seed), seed)
        except Exception:                                  # noqa: BLE001
            bad.append(seed)
            print('FAIL seed', seed)
            traceback.print_exc(limit=3)
        if seed % 20 == 19:
            print(f'{seed + 1} seeds, {len(bad)} failures', flush=True)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
