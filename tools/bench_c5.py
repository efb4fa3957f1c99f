#!/usr/bin/env python3
"""BASELINE config 5: retrieval inner loop on sampled cross sections -- 1e4 pyrat.eval() calls at
1e5 wavenumbers x 80 layers, batched walkers (bench.py --workload c5 [--gpus N]).

One step = one batch of BATCH walkers through interp_ec -> transit_path -> optical depth +
transmission -> band integration (TableSpectrum.eval_bands: one launch per stage per batch).
With N > 1 the walkers of every batch are dealt to the ranks (replicas of the table)
and the band fluxes are all-gathered.  Synthetic inputs: S=4 species, 10 table
temperatures 300-3000 K, cross sections 10**U(-30,-20) smooth in T, parameter vectors within
+-10 % of a base T / abundance vector, seed 7, per-walker hydrostatic radius.

CPU legs (reference interp_ec + optdepth + trapezoid2D from oracle/_ref, else the oracle's port):
one core on a few evals, and all cores with one eval per worker at a time (walkers are
independent, the way the reference's sampler parallelises them).  Workers start before the GPU
is touched.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BATCH = 64
# walkers per launch inside a batch (engine.TableSpectrum.eval_bands)
CHUNK = int(os.environ.get('PB_C5_CHUNK', '64'))
NSPEC, NTEMP, NLAYERS, NWAVE = 4, 10, 80, 100001
HBM_PEAK_GBS = 8000.0


def inputs(nwave=NWAVE, nlayers=NLAYERS):
    from pyratbay_amd import synth
    rng = np.random.default_rng(7)
    grid = synth.spectral_grid(4000.0, 4000.0 + (nwave - 1) * 0.05 + 0.025, 0.05, 180)
    atm = synth.synthetic_atmosphere(
        nlayers, ('H2', 'He', 'H2O', 'CO', 'CO2', 'CH4'), (0.85, 0.149, 4e-4, 5e-4, 1e-7, 1e-4))
    ttable = np.linspace(300.0, 3000.0, NTEMP)
    # smooth in T, rough in wavenumber, rising with pressure
    base = rng.uniform(-30.0, -20.0, (NSPEC, 1, 1, grid['nwave']))
    slope = rng.uniform(-1.0, 1.0, (NSPEC, 1, 1, 1)) * (ttable[None, :, None, None] / 3000.0)
    etable = 10.0**np.clip(base + slope + 0.1 * np.log10(atm['press'])[None, None, :, None],
                           -30.0, -20.0)
    nb = 24
    edges = np.linspace(100, grid['nwave'] - 100, nb + 1).astype(int)
    bands = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        resp = np.ones(hi - lo)
        bands.append((int(lo), resp, 1.0 / np.trapezoid(resp, grid['wn'][lo:hi])))
    return dict(grid=grid, atm=atm, ttable=ttable, etable=etable, bands=bands)


def walkers(inp, n, seed):
    """temps[n, L], dens[n, L, S], radius[n, L] within +-10 % of the base model."""
    rng = np.random.default_rng(seed)
    atm = inp['atm']
    temps = atm['temp'][None, :] * (1.0 + 0.1 * rng.uniform(-1, 1, (n, 1)))
    scale = 10.0**(0.0414 * rng.uniform(-1, 1, (n, 1, NSPEC)))          # +-10 % abundances
    dens = atm['dens'][None, :, 2:2 + NSPEC] * scale * (atm['temp'][None, :, None] / temps[:, :, None])
    radius = atm['radius'][None, :] * (1.0 + 0.02 * (temps[:, :1] / atm['temp'][0] - 1.0))
    return temps, dens, radius


# ------------------------------------------------------------------ CPU reference legs
def cpu_eval(arr, temps, dens, radius, rstar, bands, wn, rt='transit'):
    from oracle import oracle as orc, ref
    L, W = temps.shape[0], wn.shape[0]
    ec = np.zeros((L, W))
    if rt == 'emission':
        # interp_ec + plane_parallel_optical_depth + blackbody_wn_2D + intensity + the quadrature
        # sum (pyrat/spectrum.py:366-377, spectrum/radiative_transfer.py:76-138), default raygrid
        from pyratbay_amd import engine
        mu, weights = engine.default_quadrature()
        depth = np.zeros((L, W))
        ideep = np.full(W, L - 1, np.intc)
        if ref.available():
            kind = 'reference'
            ref.module('_extcoeff').interp_ec(ec, arr['etable'], arr['ttable'], temps, dens, 0, L)
            t = ref.module('_trapezoid')
            ideep64 = np.full(W, L - 1, int)
            t.plane_parallel_optical_depth(depth, ideep64, ec, -np.ediff1d(radius), 10.0, 0, L)
            B = np.zeros((L, W))
            ref.module('_blackbody').blackbody_wn_2D(np.ascontiguousarray(wn), temps, B, ideep64)
            inten = t.intensity(depth, ideep64, B, mu, 0)
        else:
            kind = 'port'
            orc.interp_ec(ec, arr['etable'], arr['ttable'], temps, dens, 0, L)
            orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(radius), 10.0, 0, L)
            B = orc.blackbody_wn_2D(np.ascontiguousarray(wn), temps)
            inten = orc.intensity(depth, ideep, B, mu, 0)
        spec = np.sum(inten * weights[:, None], axis=0)
        flux = np.array([np.trapezoid(spec[s:s + len(r)] * r, wn[s:s + len(r)]) * h
                         for s, r, h in bands])
        return kind, flux
    if ref.available():
        kind = 'reference'
        ref.module('_extcoeff').interp_ec(ec, arr['etable'], arr['ttable'], temps, dens, 0, L)
        t = ref.module('_trapezoid')
        raypath = orc.transit_path(radius, 0)
        depth = np.zeros_like(ec)
        ideep = np.full(W, -1, np.intc)
        for r in range(L):
            depth[r] = t.optdepth(ec[0:r + 1], raypath[r], 10.0, ideep, r)
        ideep[ideep < 0] = L - 1
        integ = np.exp(-depth) * np.expand_dims(radius, 1)
        spec = t.trapezoid2D(integ, np.ediff1d(radius), ideep.astype(np.intc))
        spec = (radius[0]**2 + 2 * spec) / rstar**2
    else:
        kind = 'port'
        orc.interp_ec(ec, arr['etable'], arr['ttable'], temps, dens, 0, L)
        depth, ideep = orc.optical_depth_transit(ec, radius, 0, L, 10.0)
        spec = orc.transmission(depth, radius, rstar, ideep, 0)
    flux = np.array([np.trapezoid(spec[s:s + len(r)] * r, wn[s:s + len(r)]) * h
                     for s, r, h in bands])
    return kind, flux


def cpu_worker():
    for line in sys.stdin:
        job = json.loads(line)
        if job.get('quit'):
            break
        d = job['dir']
        arr = {k: np.load(os.path.join(d, k + '.npy'), mmap_mode='r')
               for k in ('etable', 'ttable', 'temps', 'dens', 'radius', 'wn')}
        bands = [(int(s), np.ones(int(c)), float(h)) for s, c, h in job['bands']]
        t0 = time.perf_counter()
        out = []
        for w in job['walkers']:
            kind, flux = cpu_eval(arr, np.array(arr['temps'][w]), np.array(arr['dens'][w]),
                                  np.array(arr['radius'][w]), job['rstar'], bands, arr['wn'],
                                  job.get('rt', 'transit'))
            out.append(flux.tolist())
        print(json.dumps({'kind': kind, 'wall': time.perf_counter() - t0, 'flux': out}),
              flush=True)


def main(args):
    if getattr(args, 'cpu_worker', False):
        return cpu_worker()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    want_cpu = world == 1 and not args.no_cpu_baseline
    procs = []
    if want_cpu:
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        nworkers = max(1, min(ncores - 1, 32))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker'],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                  cwd=ROOT) for _ in range(nworkers)]

    import torch
    import torch.distributed as dist
    from pyratbay_amd import engine
    from pyratbay_amd.dist import walker_slice, gather_walkers
    rehearse = os.environ.get('PB_REHEARSE') == '1'
    torch.cuda.set_device(0 if rehearse else local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
    inp = inputs()
    g, atm = inp['grid'], inp['atm']
    nwave, nlayers = g['nwave'], atm['nlayers']
    t0 = time.perf_counter()
    rt = 'emission' if getattr(args, 'workload', 'c5') == 'c5-emission' else 'transit'
    model = engine.TableSpectrum(inp['etable'], inp['ttable'], g['wn'], atm['radius'],
                                 atm['rstar'], rt_path=rt)
    pb = engine.PassBands(g['wn'], inp['bands'])
    steps = args.steps if args.steps not in (None, 20) else 157   # default: 1e4 evals in batches of 64
    nbatch_distinct = 4
    batches = []
    for b in range(nbatch_distinct):
        temps, dens, radius = walkers(inp, BATCH, 700 + b)
        a, e = walker_slice(BATCH, world, rank)
        batches.append(tuple(engine.dev(x[a:e]) for x in (temps, dens, radius)))
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0

    def step(i):
        temps, dens, radius = batches[i % nbatch_distinct]
        flux = model.eval_bands(temps, dens, pb, radius=radius, chunk=CHUNK)
        return gather_walkers(flux, BATCH, world, rank) if world > 1 else flux

    from tools.gpu_state import Sampler
    state = Sampler()                    # (before the warm-up: see gpu_state.py)
    for i in range(args.warmup):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    with state:
        t0 = time.perf_counter()
        for i in range(steps):
            out = step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearse else 'cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the same batches through the one-pass form (model.one_pass: interpolation + optical depth +
    # transmission in one kernel, ec never stored; two walkers per wavefront where the batch allows)
    # -- reported beside `value`, which stays the default path's
    one_pass = None
    from pyratbay_amd import _capi
    if world == 1 and rt == 'transit' and _capi.experiments():
        # (experiments build of the library only: PB_LIBPBHIP=pyratbay_amd/libpbhip_exp.so)
        default_flux = out.clone()
        model.one_pass = True
        n1p = max(8, min(steps, 40))
        for i in range(2):
            out1 = step(steps - 1 + 0 * i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n1p):
            out1 = step(steps - 1 - (n1p - 1) + i)
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t1
        model.one_pass = None
        one_pass = {'evals_per_s': n1p * BATCH / el1, 'ms_per_batch': 1e3 * el1 / n1p,
                    'batches': n1p, 'ec_buffer_bytes_saved': 8 * BATCH * nlayers * nwave,
                    'bandflux_max_rel_vs_default':
                        float(torch.max(torch.abs(out1 / default_flux - 1)).item())}

    if rank == 0:
        evals = steps * BATCH
        value = evals / elapsed
        # the two big kernels of a batch, each timed alone with events on the launch stream; the
        # dominant one (by time) is the roofline block, the other is listed beside it
        temps, dens, radius = batches[0]
        nloc = temps.shape[0]
        reps = 10

        def timed(fn):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / reps

        # (as eval_bands runs them: the columns in the depth order of the first walker it saw)
        ordered = model.column_order is not None
        table = model.etable_ordered if ordered else model.etable
        # (transit geometry with ordered columns: the interpolation writes only the layers a block
        # of columns can need, TableSpectrum.tile_limit; the kernels are timed as eval_bands runs them)
        tile_limit = getattr(model, 'tile_limit', None) if rt == 'transit' else None
        interp_ms = timed(lambda: engine.interp_ec_batch(table, model.ttable, temps, dens,
                                                         tile_limit=tile_limit, row0=0))
        ec = engine.interp_ec_batch(table, model.ttable, temps, dens)
        grid_order_ms = None
        flagged = 0
        if rt == 'emission':
            intervals = (radius[:, :-1] - radius[:, 1:]).contiguous()
            transit_ms = timed(lambda: engine.emission_flux_batch(
                ec, intervals, model.wn_ordered if ordered else model.wn, temps, model.mu,
                model.weights, 0, nlayers, 10.0, model.column_order))
            depth_rows = None
        else:
            path = engine.transit_path_device(radius, 0)
            if ordered:
                tflags = torch.zeros(nloc + 1, dtype=torch.int32, device='cuda')
                transit_ms = timed(lambda: engine.transit_spectrum_ordered(
                    ec, path, radius, model.column_order, atm['rstar'], 0, nlayers, 10.0,
                    tile_limit=tile_limit, flags=tflags if tile_limit is not None else None))
                flagged = int(tflags[:nloc].sum().item())
            else:
                transit_ms = timed(lambda: engine.transit_spectrum_batch(
                    ec, path, radius, atm['rstar'], 0, nlayers, 10.0))
            # rows of ec the reference's loop reads: down to each column's first crossing
            _, _, ideep = engine.transit_spectrum_batch(ec, path, radius, atm['rstar'], 0, nlayers,
                                                        10.0, want_depth=True)
            depth_rows = float((ideep.double() + 1).sum().item())
            del ideep
        if ordered:
            ec_grid = engine.interp_ec_batch(model.etable, model.ttable, temps, dens)
            if rt == 'emission':
                grid_order_ms = timed(lambda: engine.emission_flux_batch(
                    ec_grid, intervals, model.wn, temps, model.mu, model.weights, 0, nlayers, 10.0))
            else:
                grid_order_ms = timed(lambda: engine.transit_spectrum_batch(
                    ec_grid, path, radius, atm['rstar'], 0, nlayers, 10.0))
            del ec_grid
        # compulsory bytes of ONE batched launch.  interp: every table slice (species x layer x
        # temperature node) that some walker of the batch brackets is read once -- walkers in the
        # same bracket share it -- and every walker's ec is written; transit: every walker's ec
        # is read once, its spectrum written.
        tt = np.asarray(inp['ttable'])
        th = temps.cpu().numpy()
        tlo = np.clip(np.searchsorted(tt, th, 'right') - 1, 0, len(tt) - 2)      # [nloc, L]
        nodes = sum(len(np.union1d(tlo[:, k], tlo[:, k] + 1)) for k in range(nlayers))
        interp_bytes = 8.0 * NSPEC * nwave * nodes + 8.0 * nlayers * nwave * nloc
        written_frac = 1.0
        if tile_limit is not None:
            # layers written per block of 256 ordered columns: 16 (tile + 1), table slices likewise
            lay = np.minimum(16 * (tile_limit.cpu().numpy().astype(np.int64) + 1), nlayers)
            cols = np.minimum(256, nwave - 256 * np.arange(len(lay)))
            written_frac = float((lay * cols).sum()) / (nlayers * nwave)
            interp_bytes *= written_frac
        transit_bytes = (8.0 * nlayers * nwave + 8.0 * nwave) * nloc
        if depth_rows is not None:
            # (transit: the rows down to every column's first crossing -- what the reference's
            # optical-depth loop touches, _trapezoid.c:259-273 -- not the whole of ec)
            transit_bytes = 8.0 * depth_rows + 8.0 * nwave * nloc
        mfma = os.environ.get('PB_TRANSIT_MFMA', '1') != '0'
        nmu = len(model.mu) if rt == 'emission' else 0
        kernels = [
            {'kernel': 'k_emission_fused', 'kernel_ms': transit_ms, 'kernel_bytes': transit_bytes,
             'bound_by': (f'FP64 vector ALU: per (walker, column, layer) one Planck exp + {nmu} '
                          'exp(-depth / mu) (plane_parallel_optical_depth + blackbody + intensity '
                          'in one pass, depth and B never stored); ec read once')}
            if rt == 'emission' else
            {'kernel': 'k_transit_mfma_rows<5,3,256>' if mfma else 'k_transit_pair<16>',
             'kernel_ms': transit_ms, 'kernel_bytes': transit_bytes,
             'bound_by': ('FP64 matrix pipe (at most 120 v_mfma_f64_16x16x4_f64 per 32 columns, '
                          'row tile by row tile until the 32 columns have all crossed maxdepth) + '
                          'one exp per (column, row above the crossing) on the vector ALU; the '
                          'layers of ec down to the exit tile read once') if mfma else
                         'FP64 vector ALU (3160 fma + 80 exp per column), not HBM'},
            {'kernel': 'k_interp_ec_batch2<4,true>' if os.environ.get('PB_INTERP_PAIRS', '1') != '0' else 'k_interp_ec_batch<4,true>', 'kernel_ms': interp_ms,
             'kernel_bytes': interp_bytes, 'bound_by': 'HBM (ec written per walker)'}]
        for k in kernels:
            k['achieved'] = k['kernel_bytes'] / (k['kernel_ms'] * 1e-3) / 1e9
            k['frac'] = k['achieved'] / HBM_PEAK_GBS
        kernels.sort(key=lambda k: -k['kernel_ms'])
        dom = kernels[0]
        # HBM traffic of the dominant kernel from the PMC passes (tools/pmc_traffic.py), accepted
        # only for THIS build of the library
        traffic = None
        try:
            import hashlib
            rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            so = os.path.join(ROOT, 'pyratbay_amd', 'libpbhip.so')
            if rec.get('libpbhip_sha256') == hashlib.sha256(open(so, 'rb').read()).hexdigest():
                wl = 'c5-emission' if rt == 'emission' else 'c5'
                for name, e in rec.get(wl, {}).get('kernels', {}).items():
                    if dom['kernel'].split('<')[0] in name:
                        traffic = e['hbm_bytes_per_launch']
        except Exception:                                       # noqa: BLE001
            traffic = None
        path_bytes = (16.0 * NSPEC + 32.0) * nlayers * nwave + 8.0 * nwave
        out_json = {
            'metric': 'pyrat.eval() calls/sec (1e5 wavenumbers x 80 layers, sampled cross sections'
                      + (', emission geometry)' if rt == 'emission' else ')'),
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'MCMC retrieval inner loop: {evals} eval() calls at 1e5 '
                                   'wavenumbers x 80 layers, batched walkers'
                                   + (', emission geometry (plane-parallel RT, 5-angle quadrature)'
                                      if rt == 'emission' else ''),
                       'batch': BATCH, 'nspec': NSPEC, 'ntemp': NTEMP, 'nbands': len(inp['bands']),
                       'table_bytes': int(model.etable.numel() * 8),
                       'parallelism': 'single GPU' if world == 1 else
                       f'walker replicas x{world} + all-gather of band fluxes',
                       'init_seconds': round(t_init, 3), 'one_pass': one_pass,
                       'column_order': ('depth order of the first walker (TableSpectrum.'
                                        'order_columns): spectra do not depend on it'
                                        if ordered else 'grid order'),
                       'rt_kernel_ms_grid_order': grid_order_ms,
                       'layers_written': (None if tile_limit is None else
                                          {'fraction_of_ec': written_frac,
                                           'margin_layers': model.tile_margin,
                                           'walkers_flagged_for_repair': flagged}),
                       'box_reference': box_reference(),
                       'gpu_state': state.summary()},
            'roofline': {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': dom['achieved'],
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dom['frac'],
                         'traffic': traffic, 'kernel_ms': dom['kernel_ms'],
                         'kernel_bytes': dom['kernel_bytes'], 'binding': dom['bound_by'],
                         'other_kernels': kernels[1:],
                         'note': 'bytes = what one launch of 64 walkers must move: table slices '
                                 'shared by the walkers that bracket them counted once',
                         'path_bytes_per_eval_unshared': path_bytes},
        }
        if want_cpu:
            out_json.update(cpu_legs(inp, procs, step(0).cpu().numpy(), rt))
        print(json.dumps(out_json), flush=True)
    for p in procs:
        try:
            p.stdin.write(json.dumps({'quit': True}) + '\n')
            p.stdin.flush()
            p.wait(timeout=10)
        except Exception:
            p.kill()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def box_reference():
    """What THIS box's HBM does on a plain stream, beside the bench line: fill (write-only) and
    copy (read + write) of a 4-GiB buffer through torch, GB/s.  Box-to-box spreads of the C5 line
    (26e3 ... 31e3 evals/s at equal sclk / mclk / fclk) follow these numbers."""
    import torch
    n = 1 << 29                                            # 4 GiB of doubles
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    b = torch.empty(n, dtype=torch.float64, device='cuda')
    out = {}
    for name, fn, nbytes in (('fill_GBps', lambda: a.fill_(1.0), 8.0 * n),
                             ('copy_GBps', lambda: b.copy_(a), 16.0 * n)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(5 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b
    return out


def cpu_legs(inp, procs, gpu_flux_last, rt='transit'):
    g, atm = inp['grid'], inp['atm']
    n1 = 8
    nall = max(len(procs), 1) * 2
    temps, dens, radius = walkers(inp, max(n1, nall), 700)     # = the GPU's first batch
    arr = dict(etable=inp['etable'], ttable=inp['ttable'])
    t0 = time.perf_counter()
    flux1 = []
    for w in range(n1):
        kind, f = cpu_eval(arr, temps[w], dens[w], radius[w], atm['rstar'], inp['bands'], g['wn'],
                           rt)
        flux1.append(f)
    s1 = (time.perf_counter() - t0) / n1
    res = {'cpu_baseline': dict(value=1.0 / s1, unit='evals/s', cores=1, kind=kind,
                                seconds_per_eval=s1,
                                sample=f'{n1} eval() calls ({rt}): interp_ec + optical depth + '
                                       'trapezoid2D + band trapezoids, one core')}
    # parity of the GPU batch with the CPU path on the same walkers (the first batch)
    res['cpu_baseline']['parity'] = {
        'bandflux_max_rel_err': float(np.max(np.abs(gpu_flux_last[:n1] / np.array(flux1) - 1))),
        'walkers_compared': n1}
    base = '/dev/shm' if os.path.isdir('/dev/shm') else None
    d = tempfile.mkdtemp(prefix='pb_c5_', dir=base)
    try:
        for k, v in (('etable', inp['etable']), ('ttable', inp['ttable']), ('temps', temps),
                     ('dens', dens), ('radius', radius), ('wn', g['wn'])):
            np.save(os.path.join(d, k + '.npy'), np.ascontiguousarray(v))
        bands = [(s, len(r), h) for s, r, h in inp['bands']]
        shares = [list(range(i, nall, len(procs))) for i in range(len(procs))]
        t0 = time.perf_counter()
        for p, ws in zip(procs, shares):
            p.stdin.write(json.dumps({'dir': d, 'walkers': ws, 'bands': bands, 'rt': rt,
                                      'rstar': float(atm['rstar'])}) + '\n')
            p.stdin.flush()
        replies = [json.loads(p.stdout.readline()) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        shutil.rmtree(d, ignore_errors=True)
    res['cpu_baseline_allcores'] = dict(
        value=nall / wall, unit='evals/s', cores=len(procs), kind=replies[0]['kind'],
        sample=f'{nall} eval() calls over {len(procs)} worker processes (wall {wall:.2f} s)')
    return res


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu-worker', action='store_true')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    main(ap.parse_args())
