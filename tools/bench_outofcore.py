"""Out-of-core line lists: a synthetic list of N lines (default 1e8) on C2's grid (1e5 wavenumbers
x 80 layers) under a fixed record budget (default 16 GiB), walked in chunks
(pb_lbl_set_record_budget).  Reports set-up and per-call times and, with --check, compares with
the same list in ONE un-chunked call, bit for bit (the records of 1e8 lines still fit in 288 GB;
the budget is for smaller cards and longer lists).
usage: python tools/bench_outofcore.py [--lines 1e8] [--budget-gib 16] [--check] [--out file.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lines', type=float, default=1e8)
    ap.add_argument('--budget-gib', type=float, default=16.0)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import torch
    from pyratbay_amd import engine, synth
    n = int(args.lines)
    case = synth.lbl_case(100001, 80, 1000, wnstep=0.05, niso=1, seed=42)      # grid, atmosphere
    g, atm, iso, vg = case['grid'], case['atm'], case['iso'], case['voigt']
    t0 = time.perf_counter()
    rng = np.random.default_rng(1234)
    lwn = rng.uniform(g['wn'][0], g['wn'][-1], n)
    lwn.sort()
    elow = rng.uniform(0.0, 8000.0, n)
    gf = 10.0**rng.uniform(-12.0, -6.0, n)
    lid = np.zeros(n, np.int32)
    t_gen = time.perf_counter() - t0
    vt = engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'],
                                 g['wnosamp'])
    t0 = time.perf_counter()
    ll = engine.LineList(lwn, elow, gf, lid, 1, g['own'])
    t_lines = time.perf_counter() - t0

    def plan(lines):
        return engine.LBL(vt, lines, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                          iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'],
                          vg['cutoff'], 1e-30, max_layers=80)
    t0 = time.perf_counter()
    lbl = plan(ll)
    t_plan = time.perf_counter() - t0
    budget = int(args.budget_gib * 2**30)
    lbl.set_record_budget(budget)
    t, d, z = engine.dev(atm['temp']), engine.dev(atm['dens']), engine.dev(iso['isoz'])
    ec = torch.empty((80, 1, g['nwave']), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lbl.extinction(t, d, z, add=True, out=ec)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    lbl.extinction(t, d, z, add=True, out=ec)
    torch.cuda.synchronize()
    t_again = time.perf_counter() - t0
    rec = dict(lines=n, groups=int(ll.ngroups), coadded=int(ll.nadd),
               records_bytes_whole_list=int(ll.ngroups) * 80 * 16, record_budget_bytes=budget,
               chunks=lbl.last_chunks, kernel=lbl.last_gather_kernel,
               seconds=dict(generate_lines=t_gen, pb_lines_create=t_lines, pb_lbl_create=t_plan,
                            extinction_first_call=t_first, extinction=t_again),
               device_memory_gib=torch.cuda.mem_get_info()[1] / 2**30 -
               torch.cuda.mem_get_info()[0] / 2**30)
    print(json.dumps(rec), flush=True)
    if args.check and lbl.last_chunks == 0:
        print('the records fit the budget: the call was not chunked, nothing to compare '
              '(lower --budget-gib or raise --lines)', flush=True)
    elif args.check:
        # the same list in ONE call (its records fit in HBM here: that is what the budget is
        # for on a smaller card or a longer list), one workgroup per tile as in the chunked form
        os.environ['PB_STAGE_SPLIT'] = '1'
        whole = plan(ll)
        t0 = time.perf_counter()
        want = whole.extinction(t, d, z, add=True)
        torch.cuda.synchronize()
        rec['seconds']['extinction_one_call_first'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        want = whole.extinction(t, d, z, add=True)
        torch.cuda.synchronize()
        rec['seconds']['extinction_one_call'] = time.perf_counter() - t0
        assert whole.last_chunks == 0
        rec['chunked_equals_one_call_bitwise'] = bool(torch.equal(ec, want))
        print(f'whole list in {lbl.last_chunks} chunks == one call: '
              f"{rec['chunked_equals_one_call_bitwise']}", flush=True)
        assert rec['chunked_equals_one_call_bitwise']
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(rec, f, indent=1)


if __name__ == '__main__':
    main()
