"""profiles/r02_summary.md from the bench lines under profiles/ (run after tools/profile_round.sh
and after copying its outputs into profiles/).  usage: python tools/make_summary.py [tag]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    d = {w: json.load(open(os.path.join(P, f'{tag}_bench_{w}.json'))) for w in ('c2', 'c3', 'c4', 'c5')}

    def row(w):
        x = d[w]
        r, one, many, u = x['roofline'], x['cpu_baseline'], x.get('cpu_baseline_allcores', {}), x['unit']
        return (f"| {w}: {x['config']['workload']} | {x['value']:.1f} {u} | {x['ms_per_step']:.3f} | "
                f"{r['kernel']} {r['kernel_ms']:.3f} ms | {r['frac']:.4f} | {one['value']:.4g} {u} | "
                f"{many.get('value', float('nan')):.4g} on {many.get('cores', '?')} workers |")

    stats = {}
    with open(os.path.join(P, f'{tag}_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            m = re.search(r'\b(k_[a-z_0-9]+)', r['Name'])
            if m and m.group(1) not in stats:
                stats[m.group(1)] = float(r['AverageNs']) / 1e6
    c2 = d['c2']
    r2, n = c2['roofline'], c2['north_star_target']
    b, wt = r2['binding'], r2.get('with_table', {})
    r5 = d['c5']['roofline']
    o5 = r5['other_kernels'][0]
    par = c2['cpu_baseline']['parity']
    names = ('k_ext_staged', 'k_records', 'k_transit_tau', 'k_transit_finish', 'k_layer_state',
             'k_ext_resident', 'k_path_blocks')
    step = ', '.join(f'`{k}` {stats[k]:.3f}' for k in names if k in stats)
    txt = f"""# Round 2 measurements (one MI355X, ROCm 7.2; `tools/profile_round.sh {tag}`, `tools/make_summary.py`)

Files: `{tag}_bench_{{c2,c3,c4,c5}}.json` (bench.py lines), `{tag}_kernel_stats.csv` (rocprofv3
--kernel-trace --stats of `python bench.py --no-cpu-baseline`), `pmc_traffic.json` (FETCH_SIZE /
WRITE_SIZE passes, keyed by the SHA-256 of the measured `libpbhip.so`), `{tag}_gather_ab.md` (A/B of
the gather kernels, per-layer costs, ablations and counters of `k_ext_staged`).

| workload | value | ms/step | dominant kernel (event-timed) | HBM frac | reference, 1 core | reference, all cores |
|---|---|---|---|---|---|---|
{row('c2')}
{row('c3')}
{row('c4')}
{row('c5')}

C2 step by kernel (rocprofv3 averages, ms): {step}; HIP events of bench.py give
{r2['kernel_ms']:.3f} ms for the gather.  Binding block: {b['fma_lanes_useful']:.3e} profile samples multiplied,
{b['fma_lanes_issued']:.3e} lanes issued, {b['achieved_TBps']:.1f} TB/s of LDS reads = {100 * b['frac']:.0f} % of 150 TB/s,
{b['f64_fma_TFLOPs']:.1f} TFLOP/s of useful FP64 (C3: {100 * d['c3']['roofline']['binding']['frac']:.0f} % of the LDS read rate, C4: {100 * d['c4']['roofline']['binding']['frac']:.0f} %).
Memory side: PMC traffic 2*FETCH_SIZE + WRITE_SIZE = {r2['traffic'] / 1e9:.3f} GB per launch (Infinity-Cache hits
included) against {r2['kernel_bytes'] / 1e6:.1f} MB of lines + ec (SURVEY 8d) + {r2.get('table_bytes_unique', 0) / 1e9:.3f} GB of distinct Voigt-table
samples the launch's records select (counted on the device) = {wt.get('bytes', 0) / 1e9:.3f} GB that any evaluation must move:
{wt.get('achieved', 0):.0f} GB/s = {100 * wt.get('frac', 0):.0f} % of HBM peak, measured traffic {r2['traffic'] / max(wt.get('bytes', 1), 1):.1f}x that floor.

C5 batch of 64 walkers: `{r5['kernel']}` {r5['kernel_ms']:.2f} ms (FP64 vector ALU bound: 3160 fma + 80 exp per
column; {100 * r5['frac']:.0f} % of HBM peak on the ec it must read), `{o5['kernel']}` {o5['kernel_ms']:.2f} ms =
{100 * o5['frac']:.0f} % of HBM peak on the bytes one batched launch must move (table slices shared by the walkers
that bracket them counted once, every walker's ec written).  Round 1: 4.55e3 evals/s.

north_star target ({n['workload']}): GPU {n['gpu_ms_per_spectrum']:.2f} ms per spectrum (gather {n['kernel_ms']:.2f} ms);
reference {n['cpu_baseline']['seconds_per_spectrum']:.1f} s on one core, {n['cpu_baseline_allcores']['seconds_per_spectrum']:.2f} s on {n['cpu_baseline_allcores']['cores']} workers
=> {n['speedup_vs_1core']:.0f}x / {n['speedup_vs_allcores']:.0f}x (bar: 50x).  Host: {n['cpu_baseline_allcores']['cpu']},
{n['cpu_baseline_allcores']['host_cores']} hardware threads visible.

Full-size parity of the same runs (`cpu_baseline.parity`): C2 ec {par['ec_max_rel_err']:.1e} over
{par['layers_compared']} layers, identical zero pattern, spectrum {par['spectrum_max_rel_err']:.1e}.

Per-rank compute of the multi-GPU decompositions on one GPU, collectives aside
(`tools/bench_rank.py`, `tools/bench_wshard.py`): layer-sharded N=8 0.237-0.242 ms (C2); wavenumber
shard N=2 / 4 / 8 in two phases (records and `k_records` threads for the shard's groups only; the
all-reduce of the maxima left out): 0.613 / 0.383 / 0.262 ms (0.669 / 0.416 / 0.301 before the window
map of the last session, 0.685 / 0.441 / 0.331 in the one-call form); 1e6 lines N=8 wavenumber
1.97 -> 1.28 ms.  Init (`tools/bench_init.py`): 1e7 lines `pb_lines_create` 0.342 -> 0.074 s (device
grouping), `pb_lbl_create` 1.498 -> 0.284 s.
"""
    open(os.path.join(P, f'{tag}_summary.md'), 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
