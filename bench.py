#!/usr/bin/env python3
"""Headline benchmark: transmission spectra per second at 1e5 wavenumbers x 80 layers
(BASELINE.json configs[1]: 1e5 synthetic lines, transit geometry).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the synthetic workload with every input
resident in HBM: LBL extinction (all layers) -> transit optical depth -> transmission
spectrum (the 'extinction' + 'odepth' + 'spectrum' stages of Pyrat.run(),
pyratbay/pyrat/pyrat_obj.py:203-214).  With N > 1 the spectrum is sharded over
wavenumber, one rank per GPU, and re-assembled with an RCCL all-gather (strong
scaling: the total work per step is fixed).  Rank 0 prints ONE JSON line.

The oracle / compiled reference is used only for the `cpu_baseline` leg (and never
inside the timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nwave, nlayers, nlines, wnstep, niso)
    'c2': dict(nwave=100001, nlayers=80, nlines=100000, wnstep=0.05, niso=1,
               label='1e5 wavenumbers x 80 layers, 1e5 synthetic lines, transit'),
    'c2-1e6': dict(nwave=100001, nlayers=80, nlines=1000000, wnstep=0.05, niso=1,
                   label='1e5 wavenumbers x 80 layers, 1e6 synthetic lines, transit'),
    'small': dict(nwave=10001, nlayers=20, nlines=10000, wnstep=0.05, niso=1,
                  label='1e4 wavenumbers x 20 layers, 1e4 synthetic lines, transit'),
    'c3': dict(nwave=1000001, nlayers=80, nlines=1000000, wnstep=0.005, niso=4,
               rt_path='emission',
               label='1e6 wavenumbers x 80 layers, 1e6-line 4-isotope list, emission'),
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(case, voigt, budget_layers=8, gpu_ec=None, gpu_spectrum=None,
                 rt_path='transit'):
    """Reference CPU path on a bounded sample: `budget_layers` of the layers through the
    unmodified reference _extcoeff.extinction (oracle/_ref; falls back to the oracle's
    C restatement), extrapolated to all layers, plus the full optical-depth and
    transmission stages.  Single thread (the reference holds the GIL)."""
    from oracle import oracle as orc, ref
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    nlayers = atm['nlayers']
    size = voigt.size.astype(np.int64)
    index = voigt.index.astype(np.int64)
    profile = voigt.flat()
    if ref.available():
        kind = 'reference'
        ext_fn = ref.module('_extcoeff').extinction
        int_t = np.int64
    else:
        kind = 'port'
        ext_fn = orc.extinction
        int_t = np.int32
    layers = np.unique(np.linspace(0, nlayers - 1, budget_layers).round().astype(int))
    ec = np.zeros((nlayers, g['nwave']))
    t_ext = 0.0
    for layer in layers:
        row = np.zeros((1, g['nwave']))
        t0 = time.perf_counter()
        ext_fn(row, profile, size.astype(int_t), index.astype(int_t), vg['lorentz'],
               vg['doppler'], g['wn'], g['own'], g['divisors'].astype(int_t),
               atm['dens'][layer], atm['mol_radius'], atm['mol_mass'],
               iso['isoimol'].astype(int_t), iso['isomass'], iso['isoratio'],
               iso['isoz'][:, layer].copy(), iso['isoiext'].astype(int_t), ln['lwn'],
               ln['elow'], ln['gf'], ln['lid'].astype(int_t), vg['cutoff'], case['ethresh'],
               atm['temp'][layer], 0, 1, 0)
        t_ext += time.perf_counter() - t0
        ec[layer] = row[0]
    # fill the layers that were not sampled so that the later stages see realistic columns
    for layer in range(nlayers):
        if layer not in layers:
            near = layers[np.argmin(np.abs(layers - layer))]
            ec[layer] = ec[near] * atm['press'][layer] / atm['press'][near]
    t0 = time.perf_counter()
    if kind == 'reference':
        t = ref.module('_trapezoid')
        raypath = orc.transit_path(atm['radius'], 0)
        depth = np.zeros_like(ec)
        ideep = np.full(g['nwave'], -1, np.intc)
        for r in range(nlayers):
            depth[r] = t.optdepth(ec[0:r + 1], raypath[r], case['maxdepth'], ideep, r)
        ideep[ideep < 0] = nlayers - 1
        h = np.ediff1d(atm['radius'])
        integ = np.exp(-depth) * np.expand_dims(atm['radius'], 1)
        spec = t.trapezoid2D(integ, h, (ideep).astype(np.intc))
        spec = (atm['radius'][0]**2 + 2 * spec) / atm['rstar']**2
    else:
        depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, nlayers,
                                                 case['maxdepth'])
        spec = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    t_rest = time.perf_counter() - t0
    seconds = t_ext * nlayers / len(layers) + t_rest
    # full-size parity evidence: the GPU's ec rows (and, when every layer was computed on
    # the CPU, the final spectrum) against the CPU reference on the same inputs
    parity = None
    if gpu_ec is not None:
        got, want = gpu_ec[layers], ec[layers]
        nz = want != 0
        parity = {'ec_max_rel_err': float(np.max(np.abs(got[nz] / want[nz] - 1))),
                  'ec_zero_pattern_equal': bool(np.array_equal(got == 0, want == 0)),
                  'layers_compared': int(len(layers))}
        if gpu_spectrum is not None and len(layers) == nlayers:
            parity['spectrum_max_rel_err'] = float(np.max(np.abs(gpu_spectrum / spec - 1)))
    return dict(value=1.0 / seconds, unit='spectra/s', cores=1, kind=kind, parity=parity,
                sample=(f'{len(layers)} of {nlayers} layers through extinction '
                        f'({t_ext:.2f} s), extrapolated x{nlayers / len(layers):.1f}; full '
                        f'optical depth + transmission ({t_rest:.2f} s); '
                        f'{seconds:.2f} s per spectrum'
                        + ('' if rt_path == 'transit' else
                           '; NB the transit depth/transmission stages were timed in place of '
                           f'the {rt_path} ones (extinction is >95 % of either)')))


def dominant_kernel(lbl, nlayers):
    """Name of the gather kernel that did the work of the last call: when the automatic mode
    also launched the resident-profile kernel, say which layers it actually took."""
    name = lbl.last_gather_kernel
    if name and name.startswith('k_ext_resident+'):
        resident, _ = lbl.last_layer_kinds(nlayers)
        n = int(resident.sum())
        if n == 0:
            return name.split('+', 1)[1]
        if n == nlayers:
            return 'k_ext_resident'
        return f'{name} ({n} of {nlayers} layers resident)'
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard', default='layers', choices=['layers', 'wavenumber'],
                    help='multi-GPU decomposition (see pyratbay_amd/dist.py)')
    ap.add_argument("--cpu-layers", type=int, default=80)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pyratbay_amd import engine, synth
    from pyratbay_amd.dist import SpectrumGather, LayerShardedTransit

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         'python -m torch.distributed.run --nproc-per-node N bench.py --gpus N')
    # PB_REHEARSE=1: every rank on GPU 0 with gloo (collectives staged through the host) --
    # only to rehearse the N>1 code path on a one-GPU box; never a benchmark setting
    rehearse = os.environ.get('PB_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))

    w = WORKLOADS[args.workload]
    case = synth.lbl_case(w['nwave'], w['nlayers'], w['nlines'], wnstep=w['wnstep'],
                          niso=w['niso'], seed=42)
    nwave, nlayers = case['grid']['nwave'], case['atm']['nlayers']
    rt_path = w.get('rt_path', 'transit')
    layer_mode = world > 1 and args.shard == 'layers' and rt_path == 'transit'
    t0 = time.perf_counter()
    if layer_mode:
        # layer-sharded extinction -> all-to-all -> wavenumber-sharded depth/RT -> all-gather
        sharded = LayerShardedTransit(case, world, rank)
        model = sharded.model
        wcount = nwave                      # the gather kernel covers the full grid ...
        nlayers_rank = len(sharded.layers)  # ... for this rank's layers
        step = sharded.step
    else:
        gather = SpectrumGather(nwave, world, rank, 'cuda')
        wbegin, wcount = gather.wbegin, gather.wcount
        nlayers_rank = nlayers
        model = engine.LBLSpectrum(case, rt_path=rt_path, wbegin=wbegin, wcount=wcount)

        def step():
            # every rank computes its wavenumber shard, then the shards are re-assembled
            # on every rank (RCCL all-gather over xGMI when world > 1)
            return gather(model.run())
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0

    # N > 1, layer-sharded: consecutive spectra are software-pipelined (the all-to-all and the
    # all-gather of spectrum i run beside the extinction of spectrum i+1; every spectrum of
    # the timed region is complete before its closing synchronisation).  PB_PIPELINE=0: one
    # spectrum at a time.
    pipelined = layer_mode and os.environ.get('PB_PIPELINE', '1') != '0'

    def run_steps(k):
        if pipelined:
            for _ in range(k):
                sharded.submit()
            return sharded.flush()
        out = None
        for _ in range(k):
            out = step()
        return [out]

    run_steps(args.warmup)
    model.lbl.timing_begin(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gather_ms, launches = model.lbl.timing_end()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if os.environ.get('PB_DUMP_SPECTRUM'):
        np.save(f"{os.environ['PB_DUMP_SPECTRUM']}.rank{rank}.npy",
                run_steps(3)[-1].cpu().numpy())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.steps / elapsed
        # dominant kernel: the extinction gather (k_ext_staged or k_ext_resample, see
        # roofline.kernel).  Algorithmic bytes per launch = the part of
        # SURVEY 8(d)'s per-spectrum figure that this kernel moves: read the line list
        # once (26 B/line), write ec once (8 B per layer x sample of the shard).
        n_lines = model.lines.nlines
        kernel_bytes = 26.0 * n_lines + 8.0 * nlayers_rank * wcount
        path_bytes = 26.0 * n_lines + 32.0 * nlayers * nwave + 8.0 * nwave
        kernel_ms = gather_ms / max(launches, 1)
        achieved = kernel_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if world == 1 and os.path.exists(tfile):     # measured for the single-GPU launch only
            try:
                traffic = json.load(open(tfile)).get(args.workload, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        out = {
            'metric': 'spectra/sec (1e5 wavenumbers x 80 layers)',
            'value': value, 'unit': 'spectra/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': w['label'], 'nwave': nwave, 'nlayers': nlayers,
                       'nlines': n_lines, 'wnosamp': case['grid']['wnosamp'],
                       'voigt_grid': 'nlor=100 ndop=50 extent=300 cutoff=25',
                       'voigt_table_bytes': int(model.voigt.device_bytes),
                       'parallelism': ('single GPU' if world == 1 else
                                       f'layer-sharded extinction x{world} + all-to-all + '
                                       'wavenumber-sharded RT + all-gather' +
                                       (', consecutive spectra pipelined' if pipelined else '')
                                       if layer_mode
                                       else f'wavenumber shards x{world} + all-gather'),
                       'init_seconds': round(t_init, 3)},
            'roofline': {'bound': 'hbm', 'kernel': dominant_kernel(model.lbl, nlayers_rank),
                         'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel_ms': kernel_ms, 'kernel_bytes': kernel_bytes,
                         'path_bytes_per_spectrum': path_bytes,
                         'path_GBps': path_bytes * value / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(
                case, model.voigt, args.cpu_layers, gpu_ec=model.ec.cpu().numpy()[:, 0],
                gpu_spectrum=model.spectrum.cpu().numpy() if rt_path == 'transit' else None,
                rt_path=rt_path)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
