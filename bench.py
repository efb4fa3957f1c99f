#!/usr/bin/env python3
"""Headline benchmark: transmission spectra per second at 1e5 wavenumbers x 80 layers
(BASELINE.json configs[1]: 1e5 synthetic lines, transit geometry).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c2-1e6|c3|c4|c5|small]

One step = one pass of the hot path over the synthetic workload with every input
resident in HBM: LBL extinction (all layers) -> transit optical depth -> transmission
spectrum (the 'extinction' + 'odepth' + 'spectrum' stages of Pyrat.run(),
pyratbay/pyrat/pyrat_obj.py:203-214).  With N > 1 the work is sharded, one rank per
GPU (see pyratbay_amd/dist.py), and the spectrum is re-assembled with an RCCL
all-gather (strong scaling: the total work per step is fixed).  Rank 0 prints ONE
JSON line.

The oracle / compiled reference is used only for the `cpu_baseline` legs, never inside a
timed GPU region.  The CPU legs run in worker processes that are started BEFORE this
process touches the GPU (a process that has initialised HIP is never forked or exec'd);
they mirror the reference's own parallel form: one process per layer over min(cores - 1,
nlayers) workers (pyrat/line_by_line.py:232-246, clamp at pyrat/argum.py:60-66).
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto this many hardware queues (default 4); the pipelines, the default stream
# and RCCL's own streams are more than four at N > 1, and two streams on one queue do not overlap.
# 16: two `resolution`-mode plans in flight have 2 x (1 + 4) streams of their own -- c2-res 275 ->
# 291 spectra/s against 8 queues, C2 unchanged (1123-1128 with 8, 16 or 24)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

C4_SPECIES = ('H2', 'He', 'H2O', 'CO', 'CO2', 'CH4')
C4_VMR = (0.85, 0.149, 4e-4, 5e-4, 1e-7, 1e-4)
WORKLOADS = {
    'c2': dict(nwave=100001, nlayers=80, nlines=100000, wnstep=0.05, niso=1,
               label='1e5 wavenumbers x 80 layers, 1e5 synthetic lines, transit'),
    'c2-1e6': dict(nwave=100001, nlayers=80, nlines=1000000, wnstep=0.05, niso=1,
                   label='1e5 wavenumbers x 80 layers, 1e6 synthetic lines, transit'),
    # the C2 problem on a constant-resolving-power grid (the reference's `resolution` mode,
    # _extcoeff.c:320-326): R = 123 300 puts ~1e5 samples between 4000 and 9000 cm-1; the fine
    # grid is the reference's default for that mode (wnstep 1.0 -> wnosamp 2520)
    'c2-res': dict(nwave=100001, nlayers=80, nlines=100000, wnstep=0.05, niso=1,
                   resolution=123300.0,
                   label='~1e5 wavenumbers at R = 123 300 x 80 layers, 1e5 synthetic lines, transit'),
    'small': dict(nwave=10001, nlayers=20, nlines=10000, wnstep=0.05, niso=1,
                  label='1e4 wavenumbers x 20 layers, 1e4 synthetic lines, transit'),
    'c3': dict(nwave=1000001, nlayers=80, nlines=1000000, wnstep=0.005, niso=4,
               rt_path='emission',
               label='1e6 wavenumbers x 80 layers, 1e6-line 4-isotope list, emission'),
    # c3 / c2 with a band-structured line list (synth.band_positions: 8 band heads per isotope,
    # peak line density 300 x the background's, duplicated positions): what tile dealing, segment
    # lengths and the staged / global choice see on a real molecular list
    'c3-bands': dict(nwave=1000001, nlayers=80, nlines=1000000, wnstep=0.005, niso=4,
                     rt_path='emission', bands=True,
                     label='1e6 wavenumbers x 80 layers, 1e6-line 4-isotope list with band heads '
                           '(300 x density contrast), emission'),
    'c2-bands': dict(nwave=100001, nlayers=80, nlines=100000, wnstep=0.05, niso=1, bands=True,
                     label='1e5 wavenumbers x 80 layers, 1e5 synthetic lines with band heads '
                           '(300 x density contrast), transit'),
    # BASELINE.json configs[3]; on one GPU this is its single-GPU form, with --gpus N the
    # wavenumber-sharded form the config names
    'c4': dict(nwave=1000001, nlayers=120, nlines=1000000, wnstep=0.005, niso=4,
               species=C4_SPECIES, vmr=C4_VMR, line_species=('H2O', 'CO', 'CO2', 'CH4'),
               label='1e6 wavenumbers x 120 layers, 4 species x 1e6 lines, transit'),
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LDS_PEAK_TBS = 150.0      # MI355X_MICROARCH.md: ds_read_b64/b128 aggregate, every CU streaming
NORTH_STAR = 'c2-1e6'     # north_star's >= 50x target configuration
# GPU-only legs of the default run: (workload, timed steps, warm-up steps) -- every other BASELINE
# configuration in its single-GPU form + the two other grid / list structures, so that the
# driver's one `python bench.py` times them all
LEGS = (('c3', 10, 2), ('c4', 5, 1), ('c5', 157, 4), ('c5-emission', 157, 4), ('c2-res', 50, 4),
        ('c2-bands', 50, 4))
LEG_TIMEOUT_S = 150.0


class stdout_to_stderr:
    """File descriptor 1 -> 2 for the duration: RCCL prints a version banner to the process's
    stdout when a communicator is created, and stdout carries exactly ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def metric_name(nwave, nlayers, w):
    """BASELINE.json's metric with the workload's own shape: 'spectra/sec (1e5 wavenumbers x 80
    layers)' for the headline configuration, '(1e6 wavenumbers x 120 layers)' for c4, ..."""
    e = int(round(np.log10(max(nwave - 1, 1))))
    n = f'1e{e}' if abs((nwave - 1) / 10.0**e - 1.0) < 0.02 else f'{nwave}'
    grid = f", R = {w['resolution']:.0f}" if w.get('resolution') else ''
    return f'spectra/sec ({n} wavenumbers x {nlayers} layers{grid})'


def make_case(w):
    from pyratbay_amd import synth
    kw = {k: w[k] for k in ('species', 'vmr', 'line_species', 'resolution', 'bands') if k in w}
    return synth.lbl_case(w['nwave'], w['nlayers'], w['nlines'], wnstep=w['wnstep'],
                          niso=w['niso'], seed=42, **kw)


# ---------------------------------------------------------------------------------------
# CPU reference legs (test infrastructure; nothing here is on the product path)
# ---------------------------------------------------------------------------------------
EXT_KEYS = ('profile', 'size', 'index', 'lorentz', 'doppler', 'wn', 'own', 'divisors', 'dens',
            'mol_radius', 'mol_mass', 'isoimol', 'isomass', 'isoratio', 'isoz', 'isoiext',
            'lwn', 'elow', 'gf', 'lid', 'temp')


def _ext_module():
    """(kind, extinction function, integer dtype) of the CPU implementation timed."""
    from oracle import ref
    if ref.available():
        return 'reference', ref.module('_extcoeff').extinction, np.int64
    from oracle import oracle as orc
    return 'port', orc.extinction, np.int32


def _run_layers(arr, layers, ec_out, scal, resolution=0):
    """_extcoeff.extinction for the given layers, one call per layer like
    pyrat/extinction.py:170-213; returns seconds per layer."""
    kind, ext_fn, int_t = _ext_module()
    ints = {k: np.ascontiguousarray(arr[k]).astype(int_t)
            for k in ('size', 'index', 'divisors', 'isoimol', 'isoiext', 'lid')}
    times = []
    for layer in layers:
        row = np.zeros((1, arr['wn'].shape[0]))
        t0 = time.perf_counter()
        ext_fn(row, arr['profile'], ints['size'], ints['index'], arr['lorentz'],
               arr['doppler'], arr['wn'], arr['own'], ints['divisors'], arr['dens'][layer],
               arr['mol_radius'], arr['mol_mass'], ints['isoimol'], arr['isomass'],
               arr['isoratio'], np.ascontiguousarray(arr['isoz'][:, layer]), ints['isoiext'],
               arr['lwn'], arr['elow'], arr['gf'], ints['lid'], scal['cutoff'],
               scal['ethresh'], float(arr['temp'][layer]), 0, 1, int(resolution))
        times.append(time.perf_counter() - t0)
        ec_out[layer] = row[0]
    return kind, times


def cpu_worker():
    """Worker process of the all-cores leg: waits on stdin for {"dir", "layers", ...} jobs.
    Never imports torch or touches the GPU."""
    for line in sys.stdin:
        job = json.loads(line)
        if job.get('quit'):
            break
        d = job['dir']
        arr = {k: np.load(os.path.join(d, k + '.npy'), mmap_mode='r') for k in EXT_KEYS}
        ec = np.load(os.path.join(d, 'ec.npy'), mmap_mode='r+')
        t0 = time.perf_counter()
        kind, times = _run_layers(arr, job['layers'], ec, job['scal'],
                                  job['scal'].get('resolution', 0))
        ec.flush()
        print(json.dumps({'kind': kind, 'times': times,
                          'wall': time.perf_counter() - t0}), flush=True)


class CpuPool:
    """min(cores - 1, nlayers) worker processes, started before any GPU call."""

    def __init__(self, nworkers):
        self.procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__),
                                        '--cpu-worker'], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, text=True, cwd=ROOT)
                      for _ in range(nworkers)]

    def run(self, d, nlayers, scal):
        """Every layer once, layers dealt round-robin (deep = slow layers spread evenly);
        returns (wall seconds, kind)."""
        n = min(len(self.procs), nlayers)
        shares = [list(range(w, nlayers, n)) for w in range(n)]
        t0 = time.perf_counter()
        for p, layers in zip(self.procs, shares):
            p.stdin.write(json.dumps({'dir': d, 'layers': layers, 'scal': scal}) + '\n')
            p.stdin.flush()
        replies = [json.loads(p.stdout.readline()) for p, _ in zip(self.procs, shares)]
        wall = time.perf_counter() - t0
        return wall, replies[0]['kind'], n

    def close(self):
        for p in self.procs:
            try:
                p.stdin.write(json.dumps({'quit': True}) + '\n')
                p.stdin.flush()
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def rest_of_path(case, ec, rt_path):
    """optical depth + spectrum stages on the CPU (reference modules when built)."""
    from oracle import oracle as orc, ref
    g, atm = case['grid'], case['atm']
    nlayers = atm['nlayers']
    t0 = time.perf_counter()
    if ref.available():
        t = ref.module('_trapezoid')
        raypath = orc.transit_path(atm['radius'], 0)
        depth = np.zeros_like(ec)
        ideep = np.full(g['nwave'], -1, np.intc)
        for r in range(nlayers):
            depth[r] = t.optdepth(ec[0:r + 1], raypath[r], case['maxdepth'], ideep, r)
        ideep[ideep < 0] = nlayers - 1
        h = np.ediff1d(atm['radius'])
        integ = np.exp(-depth) * np.expand_dims(atm['radius'], 1)
        spec = t.trapezoid2D(integ, h, ideep.astype(np.intc))
        spec = (atm['radius'][0]**2 + 2 * spec) / atm['rstar']**2
    else:
        depth, ideep = orc.optical_depth_transit(ec, atm['radius'], 0, nlayers,
                                                 case['maxdepth'])
        spec = orc.transmission(depth, atm['radius'], atm['rstar'], ideep, 0)
    return spec, time.perf_counter() - t0


def cpu_legs(case, voigt, pool, budget_layers, gpu_ec, gpu_spectrum, rt_path):
    """The reference CPU path on the same inputs: (a) one core, `budget_layers` sampled
    layers through _extcoeff.extinction, extrapolated to all layers; (b) all cores, every
    layer, one process per layer over the pool (wall clock); both + the optical-depth and
    spectrum stages on one core.  Returns (cpu_baseline, cpu_baseline_allcores)."""
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    nlayers, nwave = atm['nlayers'], g['nwave']
    arr = dict(profile=voigt.flat(), size=voigt.size, index=voigt.index,
               lorentz=vg['lorentz'], doppler=vg['doppler'], wn=g['wn'], own=g['own'],
               divisors=g['divisors'], dens=atm['dens'], mol_radius=atm['mol_radius'],
               mol_mass=atm['mol_mass'], isoimol=iso['isoimol'], isomass=iso['isomass'],
               isoratio=iso['isoratio'], isoz=iso['isoz'], isoiext=iso['isoiext'],
               lwn=ln['lwn'], elow=ln['elow'], gf=ln['gf'], lid=ln['lid'], temp=atm['temp'])
    scal = dict(cutoff=float(vg['cutoff']), ethresh=float(case['ethresh']),
                resolution=int(g.get('resolution') is not None))
    # (a) one core, in this process
    layers = np.unique(np.linspace(0, nlayers - 1, budget_layers).round().astype(int))
    ec1 = np.zeros((nlayers, nwave))
    kind, times = _run_layers(arr, [int(x) for x in layers], ec1, scal, scal['resolution'])
    t_ext1 = float(np.sum(times)) * nlayers / len(layers)
    # (b) all cores: arrays shared through files the workers map
    allcores = None
    ec = ec1
    if pool is not None and len(pool.procs) > 0:
        base = '/dev/shm' if os.path.isdir('/dev/shm') else None
        d = tempfile.mkdtemp(prefix='pb_cpu_', dir=base)
        try:
            for k in EXT_KEYS:
                np.save(os.path.join(d, k + '.npy'), np.ascontiguousarray(arr[k]))
            np.save(os.path.join(d, 'ec.npy'), np.zeros((nlayers, nwave)))
            wall, kind_all, nused = pool.run(d, nlayers, scal)
            ec = np.load(os.path.join(d, 'ec.npy'))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        allcores = dict(wall=wall, kind=kind_all, workers=nused)
    else:
        for layer in range(nlayers):           # fill so the later stages see real columns
            if layer not in layers:
                near = layers[np.argmin(np.abs(layers - layer))]
                ec[layer] = ec[near] * atm['press'][layer] / atm['press'][near]
    spec, t_rest = rest_of_path(case, ec, rt_path)
    note = ('' if rt_path == 'transit' else
            f'; NB the transit depth/transmission stages were timed in place of the {rt_path} '
            'ones (extinction is >95 % of either)')
    parity = None
    if gpu_ec is not None:
        cmp_layers = np.arange(nlayers) if allcores else layers
        got, want = gpu_ec[cmp_layers], ec[cmp_layers]
        nz = want != 0
        parity = {'ec_max_rel_err': float(np.max(np.abs(got[nz] / want[nz] - 1))),
                  'ec_zero_pattern_equal': bool(np.array_equal(got == 0, want == 0)),
                  'layers_compared': int(len(cmp_layers))}
        if gpu_spectrum is not None and allcores:
            parity['spectrum_max_rel_err'] = float(np.max(np.abs(gpu_spectrum / spec - 1)))
    s1 = t_ext1 + t_rest
    one = dict(value=1.0 / s1, unit='spectra/s', cores=1, kind=kind, parity=parity,
               seconds_per_spectrum=s1,
               sample=(f'{len(layers)} of {nlayers} layers through _extcoeff.extinction '
                       f'({np.sum(times):.2f} s), x{nlayers / len(layers):.2f} to all layers; '
                       f'full optical depth + transmission ({t_rest:.2f} s)' + note))
    many = None
    if allcores:
        sa = allcores['wall'] + t_rest
        many = dict(value=1.0 / sa, unit='spectra/s', cores=allcores['workers'],
                    kind=allcores['kind'], seconds_per_spectrum=sa, cpu=cpu_model(),
                    host_cores=host_cores(),
                    sample=(f'all {nlayers} layers, one process per layer over '
                            f'{allcores["workers"]} workers (the reference\'s ncpu fork model, '
                            f'wall {allcores["wall"]:.2f} s) + optical depth + transmission '
                            f'on one core ({t_rest:.2f} s)' + note))
    return one, many


# ---------------------------------------------------------------------------------------
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


def roofline_block(model, gather_ms, launches, nlayers_rank, wcount, nlayers, nwave, value,
                   traffic_workload):
    """The `roofline` object of a line-by-line workload, for its dominant kernel (the extinction
    gather).  What binds that kernel is the LDS read rate: every profile
    sample it multiplies is one 8-byte LDS read, zero lanes around a row included, counted on the
    device from the record windows of the last launch -- that is the headline block
    (`bound: "lds"`).  SURVEY 8(d)'s algorithmic HBM bytes of the same launch (line list read
    once, 26 B/line; ec written once, 8 B per layer x sample of the shard) over the same kernel
    time stand beside it as `hbm_algorithmic`, the PMC traffic as `traffic` (HBM bytes per
    launch).  A launch that kept no packed records (global gather, dynamic grids) has no lane
    count: its headline block is the HBM one."""
    n_lines = model.lines.nlines
    kernel_bytes = 26.0 * n_lines + 8.0 * nlayers_rank * wcount
    path_bytes = 26.0 * n_lines + 32.0 * nlayers * nwave + 8.0 * nwave
    kernel_ms = gather_ms / max(launches, 1)
    achieved = kernel_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    kernel = dominant_kernel(model.lbl, nlayers_rank)
    traffic = measured_traffic(traffic_workload) if traffic_workload else None
    hbm = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': achieved / HBM_PEAK_GBS, 'kernel_bytes': kernel_bytes,
           'path_bytes_per_spectrum': path_bytes, 'path_GBps': path_bytes * value / 1e9}
    # the operand SURVEY 8(d)'s byte count leaves out: the Voigt-table samples the launch's
    # live records select, each counted once (what any evaluation must read of `profile`)
    tsamp = model.lbl.last_table_samples()
    if tsamp is not None:
        wt = (kernel_bytes + 8.0 * tsamp) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        hbm['table_bytes_unique'] = 8.0 * tsamp
        hbm['with_table'] = {'bytes': kernel_bytes + 8.0 * tsamp, 'achieved': wt,
                             'frac': wt / HBM_PEAK_GBS,
                             'note': 'kernel_bytes + distinct table samples x 8 B: compare with '
                                     '`traffic`'}
    work = model.lbl.last_work()
    if work is not None and kernel_ms > 0:
        lds_tbps = work['fma_lanes_issued'] * 8.0 / (kernel_ms * 1e-3) / 1e12
        return {'bound': 'lds', 'kernel': kernel, 'achieved': lds_tbps, 'peak': LDS_PEAK_TBS,
                'unit': 'TB/s', 'frac': lds_tbps / LDS_PEAK_TBS, 'traffic': traffic,
                'kernel_ms': kernel_ms,
                'lds_bytes': work['fma_lanes_issued'] * 8.0,
                'fma_lanes_useful': work['fma_lanes_useful'],
                'fma_lanes_issued': work['fma_lanes_issued'],
                'f64_fma_TFLOPs': 2.0 * work['fma_lanes_useful'] / (kernel_ms * 1e-3) / 1e12,
                'hbm_algorithmic': hbm}
    return dict(hbm, kernel=kernel, traffic=traffic, kernel_ms=kernel_ms)


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/pmc_traffic.sh
    -- used only when that file was produced with THIS build of libpbhip.so (hash match);
    otherwise null: a stale constant is worse than none."""
    tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    so = os.path.join(ROOT, 'pyratbay_amd', 'libpbhip.so')
    try:
        rec = json.load(open(tfile))
        sha = hashlib.sha256(open(so, 'rb').read()).hexdigest()
        if rec.get('libpbhip_sha256') != sha:
            return None
        return rec.get(workload, {}).get('hbm_bytes_per_launch')
    except Exception:
        return None


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def self_launch(argv, nranks):
    """`python bench.py --gpus N` without a launcher: this process -- which has NOT touched
    the GPU (no HIP call, no torch.cuda query) and never will -- starts N fresh children, one
    per GPU, with the rendezvous variables torch.distributed.run would set, relays rank 0's
    output, and returns the first non-zero exit code (terminating the other ranks: a rank
    that died leaves its peers waiting in a collective).  Children are started with
    subprocess (fork + exec of a process that never initialised HIP); nothing is exec'd over a
    process that uses the GPU."""
    port = int(os.environ.get('MASTER_PORT', 0)) or free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks),
                   LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=env, cwd=os.getcwd(),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(nranks))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f'bench.py: rank {r} exited with code {code}; stopping the other ranks',
                      file=sys.stderr, flush=True)
                for q in sorted(pending):
                    procs[q].terminate()         # exactly the children started above
        time.sleep(0.05)
    return rc


def start_legs(names):
    """One idle child per leg, started BEFORE this process touches the GPU (`--leg-wait`: the
    child imports nothing and makes no GPU call until it reads a line from stdin)."""
    known = {n: (k, w) for n, k, w in LEGS}
    procs = []
    for name in names:
        if name not in known:
            raise SystemExit(f'--legs: unknown leg {name!r} (known: {sorted(known)})')
        steps, warm = known[name]
        cmd = [sys.executable, os.path.abspath(__file__), '--leg-wait', '--workload', name,
               '--steps', str(steps), '--warmup', str(warm), '--no-cpu-baseline',
               '--sustain-seconds', '0']
        procs.append((name, subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                             stderr=subprocess.DEVNULL, text=True, cwd=ROOT)))
    return procs


def run_legs(procs):
    """Release the leg children one after the other and keep the short form of each one's JSON
    line: value, ms_per_step, the dominant kernel and its roofline block.  A leg that fails or
    exceeds LEG_TIMEOUT_S is reported as such (and killed by PID); the others still run."""
    legs = {}
    for name, p in procs:
        t0 = time.perf_counter()
        try:
            stdout, _ = p.communicate('go\n', timeout=LEG_TIMEOUT_S)
            line = [ln for ln in stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            cfg = d.get('config', {})
            legs[name] = {
                'metric': d['metric'], 'value': d['value'], 'unit': d['unit'],
                'steps': d['steps'], 'warmup': d['warmup'], 'ms_per_step': d['ms_per_step'],
                'workload': cfg.get('workload'), 'parallelism': cfg.get('parallelism'),
                'init_seconds': cfg.get('init_seconds'), 'roofline': d.get('roofline'),
                'gpu_state': cfg.get('gpu_state'),
                'wall_seconds': round(time.perf_counter() - t0, 1)}
            for k in ('unpipelined_ms_per_spectrum', 'column_order', 'run_plans'):
                if cfg.get(k) is not None:
                    legs[name][k] = cfg[k]
        except Exception as e:                                   # noqa: BLE001
            p.kill()
            legs[name] = {'error': f'{type(e).__name__}: {e}'[:200],
                          'wall_seconds': round(time.perf_counter() - t0, 1)}
    return legs


def launch_selftest(mode):
    """Child body of `--selftest-launch` (tests/test_dist_gloo.py): a gloo rendezvous on the CPU
    with the variables self_launch() set, one all-reduce, rank 0 prints one JSON line.  mode
    'fail': rank 1 exits with code 3 before the rendezvous."""
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if mode == 'fail' and rank == 1:
        sys.exit(3)
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({'selftest': True, 'world': world, 'sum': float(t.item()),
                          'local_rank': int(os.environ['LOCAL_RANK']),
                          'master': f"{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}"}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def timed_steps(run_steps, steps, warmup, lbl, world, dist, sync, device_for_reduce,
                kernel_events=True):
    run_steps(warmup)
    lbl.timing_begin(steps if kernel_events else 0)
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gather_ms, launches = lbl.timing_end()
    if world > 1:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device_for_reduce)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, gather_ms, launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None,
                    help='timed steps (default 50; c5: 157 batches of 64 walkers = 1e4 evals)')
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS) + ['c5', 'c5-emission'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-north-star', action='store_true',
                    help='skip the 1e6-line leg of the default c2 run')
    ap.add_argument('--no-rank-projection', action='store_true',
                    help='skip the per-rank times of the wavenumber decomposition (one-rank RCCL '
                         'group) of the default c2 run')
    ap.add_argument('--shard', default='wavenumber', choices=['layers', 'wavenumber'],
                    help='multi-GPU decomposition `value` is taken from (pyratbay_amd/dist.py): '
                         "wavenumber = north_star's (shards + RCCL all-gather); the other one is "
                         'timed in the same run and reported in config.decompositions')
    ap.add_argument('--cpu-layers', type=int, default=None,
                    help='layers of the one-core CPU leg (default: all at c2, 16 otherwise)')
    ap.add_argument('--cpu-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--rank-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--selftest-launch', default=None, choices=['ok', 'fail'],
                    help=argparse.SUPPRESS)
    ap.add_argument('--no-legs', action='store_true',
                    help='skip the GPU-only legs of the default c2 run (the other BASELINE '
                         'configurations, each a child process of its own: `legs` in the JSON line)')
    ap.add_argument('--legs', default=','.join(n for n, _, _ in LEGS),
                    help='comma-separated legs of the default run (default: %(default)s)')
    ap.add_argument('--leg-wait', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--sustain-seconds', type=float, default=3.0,
                    help='length of the sustained legs at N=1 (config.sustained; 0: skip)')
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker()
    if args.rank_worker:
        return rank_worker()
    if args.leg_wait:
        # a leg of the default run: started before the parent touched the GPU, idle (nothing
        # imported, no GPU call) until the parent says go
        if not sys.stdin.readline().strip():
            return                                           # the parent went away
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the ranks ourselves (before any GPU call)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.selftest_launch:
        if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
            raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={os.environ.get("WORLD_SIZE")}')
        return launch_selftest(args.selftest_launch)
    if args.workload in ('c5', 'c5-emission'):
        from tools import bench_c5
        return bench_c5.main(args)
    if args.steps is None:
        args.steps = 50

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         'python -m torch.distributed.run --nproc-per-node N bench.py --gpus N')
    w = WORKLOADS[args.workload]
    # CPU workers first: this process has not touched the GPU yet
    pool = None
    want_cpu = world == 1 and not args.no_cpu_baseline
    if want_cpu:
        pool = CpuPool(max(1, min(host_cores() - 1, w['nlayers'])))
    rank_proc = None
    if world == 1 and args.workload == 'c2' and not args.no_rank_projection:
        # (idle until this process has finished its GPU work; see rank_worker)
        rank_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--rank-worker'],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                     cwd=ROOT)

    leg_procs = []
    if world == 1 and args.workload == 'c2' and not args.no_legs and not args.leg_wait:
        leg_procs = start_legs([n for n in args.legs.split(',') if n])

    import torch
    import torch.distributed as dist
    from pyratbay_amd import engine
    from pyratbay_amd.dist import SpectrumGather, LayerShardedTransit
    from tools.gpu_state import Sampler

    # PB_REHEARSE=1: every rank on GPU 0 with gloo (collectives staged through the host) --
    # only to rehearse the N>1 code path on a one-GPU box; never a benchmark setting
    rehearse = os.environ.get('PB_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'gloo' if rehearse else 'nccl'
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            with stdout_to_stderr():
                dist.init_process_group('nccl', rank=rank, world_size=world,
                                        device_id=torch.device('cuda', local_rank))
                # (the communicator itself is created by the first collective)
                dist.barrier()

    # the process-wide side streams are made first, in one go: HIP deals streams to hardware queues
    # in creation order, and a side stream created later (after RCCL's own streams) can land on
    # the queue of an earlier one -- two "concurrent" spectra then run one after the other
    engine.side_streams(4)
    case = make_case(w)
    nwave, nlayers = case['grid']['nwave'], case['atm']['nlayers']
    rt_path = w.get('rt_path', 'transit')
    dev_reduce = 'cpu' if rehearse else 'cuda'
    shared = {}                    # Voigt table and line list are built once per rank

    def run_decomposition(kind):
        """Build one decomposition, time it, return its numbers.  kind: 'single' (N = 1),
        'wavenumber' (north_star's: wavenumber shards + all-gather) or 'layers' (layer-sharded
        extinction + all-to-all + wavenumber-sharded RT + all-gather, consecutive spectra
        software-pipelined unless PB_PIPELINE=0)."""
        t0 = time.perf_counter()
        res = {'kind': kind}
        state = Sampler()            # (made here, entered around the timed regions: see gpu_state.py)
        per = 1                      # spectra per submission (> 1: dist.StackedShard)
        if kind == 'layers':
            sharded = LayerShardedTransit(case, world, rank, voigt=shared.get('voigt'),
                                          lines=shared.get('lines'))
            model = sharded.model
            res.update(wcount=nwave, nlayers_rank=len(sharded.layers))
            step = sharded.step
            pipelined = os.environ.get('PB_PIPELINE', '1') != '0'
        else:
            # 'replicas' (N > 1, reported beside the sharded forms, never `value`): every rank
            # computes whole, independent spectra -- what a caller after spectra/s rather than
            # after the latency of one spectrum would do with N GPUs; no data-path collective
            replicas = kind == 'replicas'
            gather = SpectrumGather(nwave, 1 if replicas else world, 0 if replicas else rank, 'cuda')
            # one GPU: consecutive spectra are independent (the temperature loop of
            # compute_opacity, the walkers of a retrieval), so `streams` of them are kept in
            # flight on as many HIP streams (engine.SpectrumPipeline; PB_STREAMS=1: one at a
            # time).  The second one fills the tail of the first one's gather launch.
            # Only where a launch has a tail worth filling: at 1e6 samples a spectrum is tens of
            # milliseconds of full-chip launches and a second one in flight gains nothing
            # (C3: 47.5 against 45.6 ms, C4: 254 against 250.5), so those run one at a time.
            # With N > 1 the wavenumber shards are pipelined the same way (dist.ShardPipeline:
            # the next spectrum's extinction beside this one's collectives; PB_STREAMS=1: off).
            streams = 2 if nwave <= 200000 else 1
            if world > 1 and not replicas and streams > 1:
                streams = 3         # a rank-size launch is short: one more hides the collectives
            streams = int(os.environ.get('PB_STREAMS', streams))
            exchange = os.environ.get('PB_KMAX_EXCHANGE', '1') != '0'
            if streams > 1 and world > 1 and not replicas:
                from pyratbay_amd.dist import ShardPipeline
                # K atmospheres per extinction call (dist.StackedShard): a rank-size gather launch
                # is short and the small launches around it do not shrink with N -- 0.159 -> 0.145
                # ms per spectrum for a 1/8 shard of C2 at K = 3 (profiles/r05_rank_stack.log);
                # PB_STACK=1: one atmosphere per call
                if rt_path == 'transit' and not w.get('resolution'):
                    per = max(1, int(os.environ.get('PB_STACK', '3')))
                # (no stage timers in the rank loop: four library calls per spectrum less for the
                # host, which at 8 ranks has ~0.18 ms per spectrum to submit everything)
                kw_pipe = dict(stack=per) if per > 1 else dict(rt_path=rt_path, timestamps=False)
                pipe = ShardPipeline(case, world, rank, depth=streams, kmax_exchange=exchange,
                                     voigt=shared.get('voigt'), lines=shared.get('lines'),
                                     **kw_pipe)
                model = pipe.models[0]
                gather = pipe.gathers[0]
                res['streams'] = streams
                res['stack'] = per
            elif streams > 1:
                pipe = engine.SpectrumPipeline(case, depth=streams, rt_path=rt_path,
                                               voigt=shared.get('voigt'),
                                               lines=shared.get('lines'))
                model = pipe.models[0]
                res['streams'] = streams
            else:
                model = engine.LBLSpectrum(case, rt_path=rt_path, wbegin=gather.wbegin,
                                           wcount=gather.wcount, voigt=shared.get('voigt'),
                                           lines=shared.get('lines'))
                if world > 1 and not replicas and exchange:
                    from pyratbay_amd.dist import kmax_allreduce
                    model.kmax_exchange = kmax_allreduce()     # records of the shard's groups only
            res.update(wcount=gather.wcount, nlayers_rank=nlayers * per)
            pipelined = False

            def step():
                # every rank computes its wavenumber shard, then the shards are re-assembled
                # on every rank (RCCL all-gather over xGMI when world > 1)
                if per > 1:
                    return pipe.gathers[0](model.run())[-1]
                return gather(model.run())
            pipelined = streams > 1
        shared.setdefault('voigt', model.voigt)
        shared.setdefault('lines', model.lines)

        def spectra_of(k):
            """Spectra that run_steps(k) computes: k rounded up to whole submissions."""
            return -(-k // per) * per

        def run_steps(k):
            if pipelined and kind != 'layers':
                # (SpectrumPipeline.submit returns this spectrum; ShardPipeline.submit the previous
                # one -- its all-gather is issued behind the next all-reduce -- and flush() the last)
                out = None
                for _ in range(-(-k // per)):
                    r = pipe.submit()
                    if r is not None:
                        out = r[0]
                last = pipe.flush()
                if last is not None:
                    out = last[0]
                return [out[-1] if isinstance(out, list) else out]
            if pipelined:
                for _ in range(k):
                    sharded.submit()
                return sharded.flush()
            out = None
            for _ in range(-(-k // per)):
                out = step()
            return [out]

        if world > 1 and kind != 'replicas':
            # parity BEFORE anything is timed: the re-assembled spectrum of this decomposition
            # (collectives included) against the same spectrum computed whole on this rank's
            # own GPU.  A rank-size launch splits the phases of a tile differently from the
            # full-grid one, so the two agree to ~1e-13, not bit for bit; the bar is 1e-10.
            if 'reference_spectrum' not in shared:
                whole = engine.LBLSpectrum(case, rt_path=rt_path, voigt=model.voigt,
                                           lines=model.lines, timestamps=False)
                shared['reference_spectrum'] = whole.run().clone()
                del whole
            got = run_steps(2)[-1]
            torch.cuda.synchronize()
            err = torch.max(torch.abs(got / shared['reference_spectrum'] - 1.0)).reshape(1)
            err = err.to(dev_reduce)
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
            res['parity_vs_single_gpu'] = float(err.item())
            if not res['parity_vs_single_gpu'] <= 1e-10:
                raise RuntimeError(f'{kind}: sharded spectrum differs from the single-GPU one '
                                   f"by {res['parity_vs_single_gpu']:.3e} (bar 1e-10)")
        if world == 1 and nwave <= 200000:
            # the same W + K steps BEFORE the priming spectra below (config.cold_value): what a
            # caller sees who runs a few dozen spectra on a chip that was idle
            el0, _, _ = timed_steps(run_steps, args.steps, args.warmup, model.lbl, world, dist,
                                    torch.cuda.synchronize, dev_reduce, kernel_events=False)
            res['cold_value'] = args.steps / el0
        if nwave <= 200000:
            # part of the set-up: a few dozen spectra so that every context's workspaces exist
            # and the chip is at its working clocks before the W warm-up steps -- with W = 5 on
            # a box that was idle the timed steps otherwise start cold (20 steps: 1020 against
            # 1075 spectra/s with a longer warm-up).  Counted in init_seconds, never timed.
            prime = int(os.environ.get('PB_PRIME', '32'))
            if kind == 'layers':
                for _ in range(prime):
                    sharded.submit() if pipelined else sharded.step()
                if pipelined:
                    sharded.flush()
            else:
                for _ in range(prime):
                    if streams > 1:
                        pipe.submit()
                    else:
                        step()
                if streams > 1:
                    pipe.flush()
            res['priming_spectra'] = prime
        torch.cuda.synchronize()
        res['init_seconds'] = round(time.perf_counter() - t0, 3)

        with state:
            elapsed, gather_ms, launches = timed_steps(
                run_steps, args.steps, args.warmup, model.lbl, world, dist,
                torch.cuda.synchronize, dev_reduce)
        res['gpu_state'] = state.summary()
        # (replicas: every rank completed `steps` spectra of its own in that time)
        nspec = spectra_of(args.steps)
        res.update(model=model, elapsed=elapsed, gather_ms=gather_ms, launches=launches,
                   value=nspec / elapsed * (world if kind == 'replicas' else 1),
                   ms_per_step=1e3 * elapsed / nspec, spectra_timed=nspec,
                   pipelined=pipelined, run_steps=run_steps)
        # the un-pipelined per-spectrum latency (one spectrum complete before the next starts)
        # beside the pipelined throughput, so that the one is not mistaken for the other
        if pipelined:
            def one_at_a_time(k):
                out = None
                for _ in range(-(-k // per)):
                    out = step()
                return [out]
            el2, g2, l2 = timed_steps(one_at_a_time, args.steps, 1, model.lbl, world, dist,
                                      torch.cuda.synchronize, dev_reduce)
            res['unpipelined_ms_per_spectrum'] = 1e3 * el2 / nspec
            if kind != 'layers':
                # the gather kernel's own duration (roofline) is the one measured with nothing
                # else on the chip, not the one stretched by the neighbouring stream
                res.update(gather_ms=g2, launches=l2)
        if world == 1 and args.sustain_seconds > 0:
            # sustained legs: >= sustain_seconds of back-to-back spectra in ONE timed region
            # (steady clocks and power state), for the pipelined form and one at a time
            def one_by_one(k):
                out = None
                for _ in range(k):
                    out = step()
                return [out]
            sus = {'seconds_target': args.sustain_seconds}
            n = max(args.steps, int(np.ceil(1.05 * args.sustain_seconds * args.steps / elapsed)))
            with state:
                el, _, _ = timed_steps(run_steps, n, 0, model.lbl, world, dist,
                                       torch.cuda.synchronize, dev_reduce, kernel_events=False)
            res['gpu_state'] = state.summary() or res.get('gpu_state')
            key = f"in_flight_{res.get('streams', 1)}"
            sus[key] = {'spectra_per_s': n / el, 'spectra': n, 'seconds': el}
            if pipelined:
                n1 = max(args.steps, int(np.ceil(1.05 * args.sustain_seconds /
                                                 (1e-3 * res['unpipelined_ms_per_spectrum']))))
                el1, _, _ = timed_steps(one_by_one, n1, 0, model.lbl, world, dist,
                                        torch.cuda.synchronize, dev_reduce, kernel_events=False)
                sus['one_at_a_time'] = {'spectra_per_s': n1 / el1, 'spectra': n1, 'seconds': el1}
            res['sustained'] = sus
        return res

    if world == 1:
        order = ['single']
    elif rt_path != 'transit':
        order = ['wavenumber']
    else:
        other = 'layers' if args.shard == 'wavenumber' else 'wavenumber'
        order = [args.shard, other, 'replicas']
    runs, failed = [], []
    for k in order:
        # a decomposition that fails (a collective the backend rejects, ...) must not take the
        # other one's number with it: it is reported in config.decompositions with its error
        try:
            runs.append(run_decomposition(k))
        except Exception as e:                                   # noqa: BLE001
            if world == 1:
                raise
            failed.append({'kind': k, 'error': f'{type(e).__name__}: {e}'[:300]})
            ok = torch.tensor([0.0], device=dev_reduce)
        else:
            ok = torch.tensor([1.0], device=dev_reduce)
        if world > 1:
            # every rank must agree on whether the decomposition ran (a rank-local failure would
            # otherwise leave the others waiting in a collective of the next one)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and runs and runs[-1]['kind'] == k:
                failed.append({'kind': k, 'error': 'failed on another rank'})
                runs.pop()
    if not runs:
        raise SystemExit(f'every decomposition failed: {failed}')
    # `value` = the --shard decomposition (default: north_star's wavenumber shards + RCCL
    # all-gather), at every N: one curve.  The other one (the extinction cut by layers) and the
    # replicas are listed beside it in config.decompositions, never `value` -- unless
    # PB_BENCH_PRIMARY=fastest asks for the faster sharded one (round 3's default).  A --shard
    # decomposition that failed leaves the other one as `value` (config.value_from says which).
    primary = runs[0]
    sharded = [r_ for r_ in runs if r_['kind'] != 'replicas']
    if not sharded:
        raise SystemExit(f'every sharded decomposition failed: {failed}')
    primary = sharded[0]
    if len(sharded) > 1 and os.environ.get('PB_BENCH_PRIMARY') == 'fastest':
        primary = max(sharded, key=lambda r_: r_['value'])
    model, elapsed = primary['model'], primary['elapsed']
    gather_ms, launches = primary['gather_ms'], primary['launches']
    wcount, nlayers_rank = primary['wcount'], primary['nlayers_rank']
    layer_mode, pipelined = primary['kind'] == 'layers', primary['pipelined']
    t_init = primary['init_seconds']
    latency_ms = primary.get('unpipelined_ms_per_spectrum')
    if os.environ.get('PB_DUMP_SPECTRUM'):
        np.save(f"{os.environ['PB_DUMP_SPECTRUM']}.rank{rank}.npy",
                primary['run_steps'](3)[-1].cpu().numpy())

    if rank == 0:
        ms_per_step = primary['ms_per_step']
        value = primary['value']
        n_lines = model.lines.nlines
        roof = roofline_block(model, gather_ms, launches, nlayers_rank, wcount, nlayers, nwave,
                              value, args.workload if world == 1 else None)
        if world == 1:
            par = 'single GPU' + (f", {primary['streams']} independent spectra in flight on "
                                  f"{primary['streams']} HIP streams" if pipelined else '')
        elif layer_mode:
            par = (f'layer-sharded extinction x{world} + all-to-all + wavenumber-sharded RT + '
                   'all-gather' + (', consecutive spectra pipelined' if pipelined else ''))
        else:
            par = (f'wavenumber shards x{world} + all-reduce(MAX) of the line-strength maxima + '
                   'all-gather' + (f", {primary['streams']} submissions of "
                                   f"{primary.get('stack', 1)} atmospheres in flight per rank"
                                   if pipelined else ''))
        out = {
            'metric': metric_name(nwave, nlayers, w),
            'value': value, 'unit': 'spectra/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': w['label'], 'nwave': nwave, 'nlayers': nlayers,
                       'nlines': n_lines, 'wnosamp': case['grid']['wnosamp'],
                       'voigt_grid': 'nlor=100 ndop=50 extent=300 cutoff=25',
                       'voigt_table_bytes': int(model.voigt.device_bytes),
                       'parallelism': par, 'backend': backend,
                       'ranks': dist.get_world_size() if world > 1 else 1,
                       'init_seconds': round(t_init, 3)},
            'roofline': roof,
        }
        if w.get('resolution'):
            # (the `resolution` path reads the layers' factors back once per call: DESIGN.md 6b)
            out['config']['run_plans'] = 'one stream synchronisation per call'
        if primary.get('spectra_timed') != args.steps:
            out['config']['spectra_timed'] = primary.get('spectra_timed')
        if latency_ms is not None:
            out['config']['unpipelined_ms_per_spectrum'] = latency_ms
        if primary.get('priming_spectra'):
            out['config']['priming_spectra_in_init'] = primary['priming_spectra']
        if primary.get('cold_value') is not None:
            out['config']['cold_value'] = primary['cold_value']
        if primary.get('sustained') is not None:
            out['config']['sustained'] = primary['sustained']
        if primary.get('gpu_state') is not None:
            # clocks / power / temperature while the sustained (else the timed) steps ran
            out['config']['gpu_state'] = primary['gpu_state']
        if world > 1:
            # both decompositions of the same run
            out['config']['value_from'] = primary['kind']
            out['config']['decompositions'] = [
                {k: r[k] for k in ('kind', 'value', 'ms_per_step', 'pipelined', 'streams', 'stack',
                                   'unpipelined_ms_per_spectrum', 'parity_vs_single_gpu',
                                   'init_seconds') if k in r}
                for r in runs] + failed
        # GPU work of this process ends here: the north_star spectrum, then the host copies the CPU
        # legs compare with; the rank-projection child gets the chip while the CPU legs run
        ns = None
        if want_cpu and args.workload == 'c2' and not args.no_north_star:
            ns = north_star_gpu(model)
        gpu_ec = model.ec.cpu().numpy()[:, 0] if want_cpu else None
        gpu_spectrum = (model.spectrum.cpu().numpy()
                        if want_cpu and rt_path == 'transit' else None)
        torch.cuda.synchronize()
        if rank_proc is not None:
            rank_proc.stdin.write('go\n')
            rank_proc.stdin.flush()
        if want_cpu:
            budget = args.cpu_layers or (nlayers if args.workload in ('c2', 'small') else 16)
            one, many = cpu_legs(case, model.voigt, pool, budget, gpu_ec=gpu_ec,
                                 gpu_spectrum=gpu_spectrum, rt_path=rt_path)
            out['cpu_baseline'] = one
            if many:
                out['cpu_baseline_allcores'] = many
            # north_star's target: >= 50x over the reference CPU _extcoeff + optical-depth
            # path on a 1e6-line / 1e5-wavenumber / 80-layer transmission spectrum at 1 GPU
            if ns is not None:
                out['north_star_target'] = north_star_leg(ns, pool)
        if rank_proc is not None:
            try:
                line, _ = rank_proc.communicate(timeout=300)
                proj = json.loads(line.strip().splitlines()[-1])
            except Exception as e:                               # noqa: BLE001
                rank_proc.kill()
                proj = {'error': f'{type(e).__name__}: {e}'[:300], 'ranks': {}}
            single = {'c2': ms_per_step,
                      NORTH_STAR: out.get('north_star_target', {}).get('gpu_ms_per_spectrum')}
            for name, rows in proj.get('ranks', {}).items():
                for row in rows.values():
                    if single.get(name):
                        # the single-GPU step (C2: two spectra in flight; 1e6 lines: one at a
                        # time) over this rank's step: the scaling available BEFORE any byte
                        # crosses xGMI
                        row['speedup_before_collectives'] = single[name] / row['ms_per_spectrum']
            out['config']['rank_projection'] = proj
        if leg_procs:
            # the other BASELINE configurations, GPU only, one child process at a time (this
            # process's GPU work is over; the CPU legs and the rank projection have finished)
            out['legs'] = run_legs(leg_procs)
        print(json.dumps(out), flush=True)
    if pool is not None:
        pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def rank_worker():
    """Child process of the default c2 run (started before the parent touches the GPU, idle until
    the parent's GPU work is over): what ONE rank of an N-GPU wavenumber-sharded run costs per
    spectrum on this one GPU -- the middle shard of N = 2, 4, 8 for C2 and for north_star's
    1e6-line list, three spectra in flight as dist.ShardPipeline runs them, both collectives issued
    through a one-rank RCCL group (the host-side cost and the stream structure of the real ones;
    no byte crosses xGMI).  A process of its own: measured inside the parent, after its pipelines,
    the same rank step took 0.23-0.26 ms instead of 0.16 (profiles/r04_summary.md)."""
    if not sys.stdin.readline().strip():
        return                                           # the parent went away
    proj = {'note': 'per-rank compute + collectives through a ONE-rank RCCL group on one GPU; '
                    'no inter-GPU traffic is measured; a rank keeps 3 submissions of '
                    '`atmospheres_per_call` atmospheres in flight (dist.StackedShard), the '
                    'single-GPU step it is compared with keeps 2 spectra in flight', 'ranks': {}}
    with stdout_to_stderr():
        try:
            import torch
            import torch.distributed as dist
            from tools import bench_rank_rccl as brr
            torch.cuda.set_device(0)
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ['MASTER_PORT'] = str(free_port())
            dist.init_process_group('nccl', rank=0, world_size=1,
                                    device_id=torch.device('cuda', 0))
            from pyratbay_amd import engine
            voigt = None
            for name in ('c2', NORTH_STAR):
                rows = {}
                stack = max(1, int(os.environ.get('PB_STACK', '3')))
                for world in (2, 4, 8):
                    # as `bench.py --gpus N` runs a rank: 3 submissions of `stack` atmospheres in
                    # flight (dist.StackedShard); and one atmosphere per call beside it
                    r = brr.measure(world, name, 3, 240, voigt=voigt, collectives=(True,),
                                    keep=True, stack=stack)
                    ms, host = r['collectives through RCCL (one rank)']
                    voigt = r['voigt']                   # same grid: one table for all of them
                    rows[f'N={world}'] = {'ms_per_spectrum': ms, 'host_submission_ms': host,
                                          'atmospheres_per_call': stack}
                    if stack > 1:
                        r1 = brr.measure(world, name, 3, 200, voigt=voigt, collectives=(True,))
                        rows[f'N={world}']['one_atmosphere_per_call_ms'] = \
                            r1['collectives through RCCL (one rank)'][0]
                proj['ranks'][name] = rows
            dist.destroy_process_group()
        except Exception as e:                           # noqa: BLE001
            proj['error'] = f'{type(e).__name__}: {e}'[:300]
    print(json.dumps(proj), flush=True)


def north_star_gpu(c2_model):
    """GPU half of the north_star leg: the 1e6-line x 1e5-wavenumber x 80-layer transit spectrum,
    step time of one spectrum at a time."""
    import torch
    from pyratbay_amd import engine
    w = WORKLOADS[NORTH_STAR]
    case = make_case(w)
    # same grid and width grids as c2: the Voigt table is shared
    model = engine.LBLSpectrum(case, rt_path='transit', voigt=c2_model.voigt)
    steps = 10
    for _ in range(2):
        model.run()
    model.lbl.timing_begin(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.run()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / steps
    gather_ms, launches = model.lbl.timing_end()
    nwave, nlayers = case['grid']['nwave'], case['atm']['nlayers']
    roof = roofline_block(model, gather_ms, launches, nlayers, nwave, nlayers, nwave, 1.0 / gpu_s,
                          NORTH_STAR)
    return dict(w=w, case=case, model=model, gpu_s=gpu_s, gather_ms=gather_ms, launches=launches,
                roofline=roof,
                ec=model.ec.cpu().numpy()[:, 0], spectrum=model.spectrum.cpu().numpy())


def north_star_leg(ns, pool):
    """CPU half: the reference on the same box (one core on 16 sampled layers, all cores on
    every layer) and the comparison."""
    w, case, model, gpu_s = ns['w'], ns['case'], ns['model'], ns['gpu_s']
    one, many = cpu_legs(case, model.voigt, pool, 16, gpu_ec=ns['ec'],
                         gpu_spectrum=ns['spectrum'], rt_path='transit')
    leg = {'workload': w['label'], 'gpu_ms_per_spectrum': 1e3 * gpu_s,
           'gpu_spectra_per_s': 1.0 / gpu_s, 'kernel': model.lbl.last_gather_kernel,
           'kernel_ms': ns['gather_ms'] / max(ns['launches'], 1), 'roofline': ns['roofline'],
           'cpu_baseline': one, 'cpu_baseline_allcores': many,
           'speedup_vs_1core': one['seconds_per_spectrum'] / gpu_s,
           'target_speedup': 50.0}
    if many:
        leg['speedup_vs_allcores'] = many['seconds_per_spectrum'] / gpu_s
    leg['target_met'] = bool(min(leg['speedup_vs_1core'],
                                 leg.get('speedup_vs_allcores', np.inf)) >= 50.0)
    return leg


if __name__ == '__main__':
    main()
