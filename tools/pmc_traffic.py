"""HBM traffic of the dominant kernel from rocprofv3 PMC passes -> profiles/pmc_traffic.json, keyed
by the hash of the libpbhip.so that was measured (bench.py refuses the file for any other build).

Run on the GPU box:   python tools/pmc_traffic.py [workload ...]
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one), with --pmc only
(never combined with tracing).  Per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes for 16-byte-per-lane loads (the row loads of the gather are
`buffer_load_dwordx4 ... lds`), so bytes = 2 * FETCH_SIZE + WRITE_SIZE; the counters are in KiB.
These are L2 memory-side requests: Infinity-Cache hits are included -- an upper bound on HBM bytes.
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_pass(counter, workload, out):
    env = dict(os.environ, TMPDIR='/tmp')
    if workload.startswith('c5'):
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '4',
               '--warmup', '1', '--no-cpu-baseline']
    else:
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'bench_stages.py'), workload, '3']
    subprocess.run(['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', out, '--'] + cmd,
                   check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
            acc.setdefault(name, []).append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def entry(kern, fetch, write):
    f_kib, w_kib = fetch.get(kern, 0.0), write.get(kern, 0.0)
    return {'kernel': kern, 'FETCH_SIZE_KiB': f_kib, 'WRITE_SIZE_KiB': w_kib,
            'hbm_bytes_per_launch': (2 * f_kib + w_kib) * 1024.0,
            'raw_fetch_plus_write_bytes': (f_kib + w_kib) * 1024.0}


def main():
    workloads = sys.argv[1:] or ['c2']
    so = os.path.join(ROOT, 'pyratbay_amd', 'libpbhip.so')
    res = {'libpbhip_sha256': hashlib.sha256(open(so, 'rb').read()).hexdigest()}
    for wl in workloads:
        base = os.path.join(ROOT, 'gpurun_out', f'pmc_traffic_{wl}')
        fetch = counter_pass('FETCH_SIZE', wl, base + '_f')
        write = counter_pass('WRITE_SIZE', wl, base + '_w')
        if wl.startswith('c5'):
            # the retrieval batch: every big kernel of a 64-walker launch (bench_c5 picks its
            # dominant one); WRITE_SIZE of 8-byte-per-lane stores is tallied double on gfx950
            # (tools/write_size_probe.hip), FETCH_SIZE of 16-byte-per-lane loads at half
            keep = [k for k in fetch if any(t in k for t in ('k_transit_mfma', 'k_interp_ec_batch',
                                                              'k_emission_fused', 'k_table_transit'))]
            res[wl] = {'kernels': {k: entry(k, fetch, write) for k in keep}}
            for k in keep:
                print(wl, res[wl]['kernels'][k])
            continue
        kern = max((k for k in fetch if 'k_ext_' in k and 'resident' not in k),
                   key=lambda k: fetch[k])
        res[wl] = entry(kern, fetch, write)
        print(wl, res[wl])
    res['note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean over the '
                   'dispatches of the dominant gather kernel; hbm_bytes_per_launch = 2*FETCH_SIZE + '
                   'WRITE_SIZE (gfx950 correction for 16-byte-per-lane loads, '
                   'MI355X_MICROARCH.md); L2 memory-side requests incl. Infinity-Cache hits')
    out = os.path.join(ROOT, 'gpurun_out', 'pmc_traffic.json')
    json.dump(res, open(out, 'w'), indent=1)
    print('wrote', out, '(copy to profiles/pmc_traffic.json)')


if __name__ == '__main__':
    main()
