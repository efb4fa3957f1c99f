"""Cycle account of one segment step of k_ext_staged.

Needs the experiments build of the library (its kProbe = 4 instantiation of the staged kernel:
valid sums, every wavefront stamps s_memtime around the parts of its work):

    make -C pyratbay_amd/csrc EXPERIMENTS=1
    PB_LIBPBHIP=$PWD/pyratbay_amd/libpbhip_exp.so python tools/step_account.py [workload] > account.md

Parent: runs itself as a child with PB_STAGE_PROBE=4 (the library prints one line per layer on
stderr), a second child without the probe for the un-instrumented kernel time, and prints the
table in markdown: per category the share of the wavefronts' lifetime, per layer class (rows of
the layer's first isotope <= 384, <= 768, longer) and over all layers, and the same in cycles per
segment step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CATS = ['isotope set-up (candidate search, scans)', 'batch: record fetch + decode',
        'batch: segment detection + table', 'find_hits (per 64 segments)',
        'step: row DMA issue', 'step: walk (hit decode, broadcasts, row reads, FMAs)',
        'step: s_waitcnt vmcnt(0) (row DMA)', 'step: s_barrier', 'batch: pipeline fill',
        'lifetime', 'steps', 'batches', 'visits', 'steps with a visit',
        'accumulator init + write-back']


def child(name, steps):
    import torch
    import bench
    from pyratbay_amd import engine
    w = bench.WORKLOADS[name]
    case = bench.make_case(w)
    model = engine.LBLSpectrum(case, rt_path=w.get('rt_path', 'transit'))
    model.lbl.set_gather_mode('staged')
    for _ in range(2):
        model.extinction()
    torch.cuda.synchronize()
    print('PROBE_BEGIN', file=sys.stderr, flush=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    model.lbl.timing_begin(steps)
    ev0.record()
    for _ in range(steps):
        model.extinction()
    ev1.record()
    torch.cuda.synchronize()
    gather_ms, launches = model.lbl.timing_end()
    work = model.lbl.last_work()
    print(json.dumps({'workload': name, 'steps': steps,
                      'extinction_ms': ev0.elapsed_time(ev1) / steps,
                      'gather_ms': gather_ms / max(launches, 1), 'kernel': model.lbl.last_gather_kernel,
                      'nlayers': model.nlayers, 'work': work}), flush=True)


def run_child(name, steps, probe):
    env = dict(os.environ)
    if probe:
        env['PB_STAGE_PROBE'] = '4'
    else:
        env.pop('PB_STAGE_PROBE', None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', name, str(steps)],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        raise SystemExit(f'child failed ({p.returncode})')
    info = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    lines = p.stderr.split('PROBE_BEGIN', 1)[-1].splitlines()
    rows = [ln.split() for ln in lines if ln.startswith('STAGE_PROBE')]
    return info, rows


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        return child(sys.argv[2], int(sys.argv[3]))
    name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    steps = 5
    plain, _ = run_child(name, steps, False)
    info, rows = run_child(name, steps, True)
    nl = info['nlayers']
    import numpy as np
    data = np.zeros((nl, 21))
    rowmax = np.zeros(nl, int)
    for r in rows:                       # STAGE_PROBE layer L rowmax R : v0 ... v14  (summed over launches)
        layer = int(r[2])
        rowmax[layer] = int(r[4])
        data[layer] += np.array([float(x) for x in r[6:27]])
    data /= steps
    # calibration from the probe itself: s_memtime ticks per 10-ns tick of the 100-MHz wall clock
    # (s_memrealtime), both taken over every wavefront's lifetime; the cost of one stamp with an
    # empty LDS queue (two back-to-back intervals per wavefront)
    ticks_per_10ns = data[:, 9].sum() / data[:, 15].sum()
    tick_ns = 10.0 / ticks_per_10ns
    ghz = float(os.environ.get('PB_SCLK_GHZ', '2.4'))
    cyc = tick_ns * ghz                                   # shader cycles per s_memtime tick
    stamp = data[:, 16].sum() / (2 * data[:, 17].sum())   # ticks per stamp
    classes = [('rows <= 384 samples', rowmax <= 384), ('rows 385 ... 768', (rowmax > 384) & (rowmax <= 768)),
               ('rows > 768', rowmax > 768), ('all layers', rowmax >= 0)]
    print(f'# Cycle account of `k_ext_staged`, workload {name}\n')
    print(f'* un-instrumented kernel: {plain["gather_ms"]:.3f} ms per launch ({plain["kernel"]}); '
          f'with the probe (stamps drain the LDS queue): {info["gather_ms"]:.3f} ms')
    print(f'* clock: {ticks_per_10ns:.3f} s_memtime ticks per 10 ns of s_memrealtime = one tick is '
          f'{tick_ns:.3f} ns = {cyc:.2f} shader cycles at {ghz} GHz.  One stamp (s_memtime + '
          f's_waitcnt lgkmcnt(0)) with an empty LDS queue costs {stamp:.1f} ticks = '
          f'{stamp * cyc:.0f} cycles; every category of a step below CONTAINS one stamp.  All '
          'figures are sums over the wavefronts of a launch, averaged over the launches.')
    w = info.get('work') or {}
    if w:
        print(f"* work of the launch: {w['live_records']} live records, "
              f"{w['fma_lanes_issued']:.4g} lane-FMAs issued ({w['fma_lanes_useful']:.4g} useful)")
    print()
    hdr = '| part of a wavefront\'s lifetime | ' + ' | '.join(c for c, _ in classes) + ' |'
    print(hdr)
    print('|---|' + '---|' * len(classes))
    data[:, 4] += data[:, 18]            # (shares: requesting and non-requesting wavefronts together)
    order = [5, 7, 6, 4, 3, 1, 2, 8, 0, 14]
    for c in order:
        cells = []
        for _, m in classes:
            life = data[m, 9].sum()
            cells.append(f'{100 * data[m, c].sum() / life:.1f} %' if life > 0 else '-')
        print(f'| {CATS[c]} | ' + ' | '.join(cells) + ' |')
    cells = []
    for _, m in classes:
        life = data[m, 9].sum()
        rest = life - data[m][:, order].sum()
        cells.append(f'{100 * rest / life:.1f} %' if life > 0 else '-')
    print('| not stamped (loop control, stamps themselves) | ' + ' | '.join(cells) + ' |')
    print()
    print('| per segment step and wavefront (one stamp subtracted from each line) | ' +
          ' | '.join(c for c, _ in classes) + ' |')
    print('|---|' + '---|' * len(classes))

    def per_step(c, m):
        st = data[m, 10].sum()
        return (data[m, c].sum() / st - stamp) * cyc if st > 0 else float('nan')
    for label, c in (('walk, cycles', 5), ('barrier wait, cycles', 7), ('DMA wait, cycles', 6),
                     ('DMA issue slot (all 8 wavefronts averaged), cycles', 4)):
        print(f'| {label} | ' + ' | '.join(f'{per_step(c, m):.0f}' for _, m in classes) + ' |')
    # the requesting wavefront against the others (whose interval holds nothing but the stamp)
    print('| ... of which: the wavefront that requests the row, cycles per request | ' +
          ' | '.join(f'{((data[m, 4].sum() - data[m, 18].sum()) / max(data[m, 19].sum(), 1) - stamp) * cyc:.0f}'
                     for _, m in classes) + ' |')
    print('| ...... its descriptor chain (LDS read of the descriptor, readfirstlanes) up to the first load | ' +
          ' | '.join(f'{(data[m, 20].sum() / max(data[m, 19].sum(), 1)) * cyc:.0f}' for _, m in classes) + ' |')
    print('| ... the other seven (a stamp right after the barrier release), cycles | ' +
          ' | '.join(f'{(data[m, 18].sum() / max(data[m, 10].sum() - data[m, 19].sum(), 1) - stamp) * cyc:.0f}'
                     for _, m in classes) + ' |')
    print('| find_hits (amortised over its 64 steps), cycles | ' +
          ' | '.join(f'{data[m, 3].sum() / max(data[m, 10].sum(), 1) * cyc:.0f}' for _, m in classes) + ' |')
    print('| whole step without the stamps, cycles | ' +
          ' | '.join(f'{sum(per_step(c, m) for c in (4, 5, 6, 7)) + data[m, 3].sum() / max(data[m, 10].sum(), 1) * cyc:.0f}'
                     for _, m in classes) + ' |')
    print('| un-instrumented: kernel time x wavefront slots / steps, cycles | ' +
          ' | '.join('-' if i < len(classes) - 1 else
                     f'{plain["gather_ms"] * 1e-3 * 256 * 32 * ghz * 1e9 / max(data[:, 10].sum(), 1):.0f} (slots 100 % busy)'
                     for i in range(len(classes))) + ' |')
    print('| layers | ' + ' | '.join(str(int(m.sum())) for _, m in classes) + ' |')
    print('| steps per (tile, wavefront) launch-wide: steps | ' +
          ' | '.join(f'{data[m, 10].sum():.3g}' for _, m in classes) + ' |')
    print('| (record, sub-tile) visits per step and wavefront | ' +
          ' | '.join(f'{data[m, 12].sum() / max(data[m, 10].sum(), 1):.2f}' for _, m in classes) + ' |')
    print('| steps in which a wavefront visits anything | ' +
          ' | '.join(f'{100 * data[m, 13].sum() / max(data[m, 10].sum(), 1):.0f} %' for _, m in classes) + ' |')
    print('| records per batch of 512 slots: batches | ' +
          ' | '.join(f'{data[m, 11].sum() / 8:.3g}' for _, m in classes) + ' |')
    print()
    print('Raw per-layer sums (ticks per launch): layer, rowmax, ' + ', '.join(str(i) for i in range(21)))
    for layer in range(nl):
        print(f'    {layer} {rowmax[layer]} ' + ' '.join(f'{v:.0f}' for v in data[layer]))


if __name__ == '__main__':
    main()
