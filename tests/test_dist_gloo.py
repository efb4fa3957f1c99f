"""The N > 1 path on CPU: two gloo ranks shard the wavenumber axis, all-gather the
spectrum and all-reduce the band fluxes (world_size 2, 127.0.0.1).

The per-shard spectra come from the oracle (test infrastructure) because the HIP kernels
need a GPU; what is under test is the product's sharding / collective code
(pyratbay_amd.dist) -- that shards computed on the GLOBAL grid concatenate to exactly the
single-rank result and that partial band integrals add up."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _bands(wn):
    out = []
    n = len(wn)
    for lo, hi in ((n // 10, 2 * n // 5), (2 * n // 5 - 3, 9 * n // 10), (0, n)):
        x = np.linspace(-1, 1, hi - lo)
        resp = np.exp(-4 * x**2)
        out.append((lo, resp, 1.0 / np.trapezoid(resp, wn[lo:hi])))
    return out


def _partial_band(spectrum, wn, bands, wbegin, wcount):
    """NumPy statement of pb_band_integrate's contract (pairs with left sample in shard)."""
    out = np.zeros(len(bands))
    for b, (start, resp, _) in enumerate(bands):
        i = np.arange(len(resp) - 1)
        g = start + i
        keep = (g >= wbegin) & (g < wbegin + wcount)
        y = spectrum[start:start + len(resp)] * resp
        out[b] = np.sum((0.5 * (wn[g + 1] - wn[g]) * (y[i] + y[i + 1]))[keep])
    return out


def _worker(rank, world, port, nwave, tmp, uniform=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    import cases
    from oracle import oracle as orc

    c = cases.column_case(seed=21, nlayers=12, nwave=nwave)
    gather = pbd.SpectrumGather(nwave, world, rank, 'cpu', uniform=uniform)
    a, n = gather.wbegin, gather.wcount
    # this rank's columns only, computed on the global grid
    depth, ideep = orc.optical_depth_transit(c['ec'][:, a:a + n].copy(), c['radius'], 0, 12,
                                             10.0)
    local = orc.transmission(depth, c['radius'], c['rstar'], ideep, 0)
    if uniform and gather.uniform:
        # the way dist.ShardPipeline uses it: the shard is written straight into the slot
        gather.slot.copy_(torch.from_numpy(local))
        full = gather(gather.slot).numpy().copy()
    else:
        full = gather(torch.from_numpy(local)).numpy().copy()
    bands = _bands(c['wn'])
    partial = torch.from_numpy(_partial_band(full, c['wn'], bands, a, n))
    heights = torch.tensor([b[2] for b in bands], dtype=torch.float64)
    bandflux = pbd.allreduce_bandflux(partial, heights).numpy()
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), full=full, bandflux=bandflux,
             bounds=gather.bounds)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nwave,world', [(1001, 2), (64, 2), (1003, 3), (1001, 8)])
def test_two_rank_shards_reassemble(tmp_path, orc, nwave, world):
    """(world 3 and 8: uneven shards and the rank count of the node the driver benches on)"""
    import cases
    mp.spawn(_worker, args=(world, _free_port(), nwave, str(tmp_path)), nprocs=world,
             join=True)
    c = cases.column_case(seed=21, nlayers=12, nwave=nwave)
    depth, ideep = orc.optical_depth_transit(c['ec'], c['radius'], 0, 12, 10.0)
    want = orc.transmission(depth, c['radius'], c['rstar'], ideep, 0)
    bands = _bands(c['wn'])
    want_flux = np.array([np.trapezoid(want[s:s + len(r)] * r, c['wn'][s:s + len(r)]) * h
                          for s, r, h in bands])
    for rank in range(world):
        got = np.load(tmp_path / f'rank{rank}.npz')
        assert np.array_equal(got['full'], want)          # bit-exact re-assembly
        np.testing.assert_allclose(got['bandflux'], want_flux, rtol=1e-13)
        assert got['bounds'][0] == 0 and got['bounds'][-1] == nwave


@pytest.mark.parametrize('nwave,world', [(1001, 2), (1003, 3), (1001, 8), (1000, 8), (10, 8)])
def test_uniform_shards_need_no_unpacking(tmp_path, orc, nwave, world):
    """SpectrumGather(uniform=True): shards of one size (the last one shorter), every rank writes
    its shard into its slot of the send buffer and the receive buffer of ONE all-gather is the
    spectrum -- no copy in, no unpacking.  (10 samples on 8 ranks: a rank would own nothing, the
    object falls back to the balanced shards.)"""
    import cases
    from pyratbay_amd.dist import uniform_bounds
    mp.spawn(_worker, args=(world, _free_port(), nwave, str(tmp_path), True), nprocs=world,
             join=True)
    c = cases.column_case(seed=21, nlayers=12, nwave=nwave)
    depth, ideep = orc.optical_depth_transit(c['ec'], c['radius'], 0, 12, 10.0)
    want = orc.transmission(depth, c['radius'], c['rstar'], ideep, 0)
    ub = uniform_bounds(nwave, world)
    assert (ub is None) == (nwave == 10)
    for rank in range(world):
        got = np.load(tmp_path / f'rank{rank}.npz')
        assert np.array_equal(got['full'], want)
        assert got['bounds'][0] == 0 and got['bounds'][-1] == nwave
        if ub is not None:
            assert np.array_equal(got['bounds'], ub)
            sizes = np.diff(ub)
            assert np.all(sizes[:-1] == sizes[0]) and 0 < sizes[-1] <= sizes[0]


def _worker_stack(rank, world, port, nwave, stack, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    gather = pbd.StackGather(nwave, world, rank, stack, 'cpu')
    a, n = gather.wbegin, gather.wcount
    want = _stack_spectra(nwave, stack)
    outs = []
    for it in range(2):                       # (buffers are re-used from call to call)
        slots = gather.slots
        for k in range(stack):
            if (k + it) % 2:                  # written in place, as StackedShard's transit calls do
                slots[k].copy_(torch.from_numpy(want[k, a:a + n] + it))
            else:                             # or handed over: copied into the send buffer
                slots[k] = torch.from_numpy(want[k, a:a + n] + it)
        outs.append(np.stack([f.numpy().copy() for f in gather(slots)]))
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), first=outs[0], second=outs[1],
             bounds=gather.bounds)
    dist.barrier()
    dist.destroy_process_group()


def _stack_spectra(nwave, stack):
    rng = np.random.default_rng(5)
    return rng.uniform(0.0, 1.0, (stack, nwave))


@pytest.mark.parametrize('nwave,world,stack', [(1001, 2, 3), (1003, 3, 2), (1000, 8, 3), (10, 8, 2)])
def test_stack_gather(tmp_path, nwave, world, stack):
    """dist.StackGather: the K shards of a stacked submission in ONE all-gather, sorted into K
    contiguous spectra on every rank (10 samples on 8 ranks: balanced shards, unpacked per rank)."""
    mp.spawn(_worker_stack, args=(world, _free_port(), nwave, stack, str(tmp_path)), nprocs=world,
             join=True)
    want = _stack_spectra(nwave, stack)
    for rank in range(world):
        got = np.load(tmp_path / f'rank{rank}.npz')
        assert np.array_equal(got['first'], want)
        assert np.array_equal(got['second'], want + 1)
        assert got['bounds'][0] == 0 and got['bounds'][-1] == nwave


def test_shard_bounds_properties():
    from pyratbay_amd.dist import shard_bounds
    for nwave in (1, 7, 8, 100001, 1000001):
        for world in (1, 2, 3, 4, 8):
            b = shard_bounds(nwave, world)
            assert b[0] == 0 and b[-1] == nwave and len(b) == world + 1
            sizes = np.diff(b)
            assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1


def _worker_layers(rank, world, port, nlayers, nwave, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    full = np.arange(nlayers * nwave, dtype=float).reshape(nlayers, nwave)   # "ec"
    lp = -(-nlayers // world)
    mine = np.zeros((lp, nwave))
    rows = np.arange(rank, nlayers, world)
    mine[:len(rows)] = full[rows]
    cols, bounds = pbd.layer_exchange(torch.from_numpy(mine), nlayers, nwave, world, rank)
    a, b = int(bounds[rank]), int(bounds[rank + 1])
    ok = np.array_equal(cols.numpy(), full[:, a:b])
    np.save(os.path.join(tmp, f'ok{rank}.npy'), np.array([ok, cols.shape[0], cols.shape[1]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nlayers,nwave,world', [(7, 101, 2), (8, 64, 2), (13, 257, 3),
                                                 (80, 1001, 8), (5, 40, 8)])
def test_layer_exchange_two_ranks(tmp_path, nlayers, nwave, world):
    """[my layers, all columns] -> [all layers, my columns] through all_to_all_single
    (also more ranks than some ranks have layers for: 5 layers on 8 ranks)."""
    mp.spawn(_worker_layers, args=(world, _free_port(), nlayers, nwave, str(tmp_path)),
             nprocs=world, join=True)
    for rank in range(world):
        ok, nl, nw = np.load(tmp_path / f'ok{rank}.npy')
        assert ok == 1 and nl == nlayers


def _worker_walkers(rank, world, port, nwalkers, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    full = np.arange(nwalkers * 3, dtype=float).reshape(nwalkers, 3)
    a, b = pbd.walker_slice(nwalkers, world, rank)
    out = pbd.gather_walkers(torch.from_numpy(full[a:b].copy()), nwalkers, world, rank)
    np.save(os.path.join(tmp, f'w{rank}.npy'), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nwalkers,world', [(11, 2), (29, 3), (64, 8), (5, 8)])
def test_walker_replicas_gather(tmp_path, nwalkers, world):
    mp.spawn(_worker_walkers, args=(world, _free_port(), nwalkers, str(tmp_path)),
             nprocs=world, join=True)
    want = np.arange(nwalkers * 3, dtype=float).reshape(nwalkers, 3)
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f'w{rank}.npy'), want)


def _worker_pipeline(rank, world, port, nlayers, nwave, nsteps, tmp):
    """ExchangePipeline over real (asynchronous) gloo collectives: produce = this rank's layers
    of a step-dependent matrix, consume = a column-wise weighted sum over all layers."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    lp = -(-nlayers // world)
    rows = np.arange(rank, nlayers, world)
    weights = torch.arange(1, nlayers + 1, dtype=torch.float64)
    state = {'step': 0}

    def matrix(step):
        return (np.arange(nlayers * nwave, dtype=float).reshape(nlayers, nwave) % 97
                + 1000.0 * step)

    def produce(ec):
        ec.zero_()
        ec[:len(rows)] = torch.from_numpy(matrix(state['step'])[rows])
        state['step'] += 1

    def consume(ec_cols):
        assert ec_cols.shape[0] == nlayers
        return (ec_cols * weights[:, None]).sum(0)

    ec = torch.zeros((lp, nwave), dtype=torch.float64)
    pipe = pbd.ExchangePipeline(ec, nlayers, world, rank, produce, consume)
    outs = []
    for _ in range(nsteps):
        o = pipe.submit()
        if o is not None:
            outs.append(o.clone().numpy())
    outs += [o.clone().numpy() for o in pipe.flush()]
    # and a second batch through the same (now empty) pipeline
    for _ in range(2):
        o = pipe.submit()
        assert o is None
    outs += [o.clone().numpy() for o in pipe.flush()]
    ok = len(outs) == nsteps + 2
    for k, got in enumerate(outs):
        want = (matrix(k) * weights.numpy()[:, None]).sum(0)
        ok = ok and np.array_equal(got, want)
    np.save(os.path.join(tmp, f'p{rank}.npy'), np.array([ok, len(outs)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nlayers,nwave,nsteps,world', [(7, 101, 5, 2), (8, 64, 1, 2), (5, 33, 2, 2),
                                                        (13, 257, 4, 3), (80, 1001, 3, 8)])
def test_exchange_pipeline_two_ranks(tmp_path, nlayers, nwave, nsteps, world):
    """The software-pipelined layer-sharded step (submit/flush: all-to-all and all-gather of
    spectrum i beside the production of i+1, double buffers) returns every spectrum, in
    order, on every rank."""
    mp.spawn(_worker_pipeline,
             args=(world, _free_port(), nlayers, nwave, nsteps, str(tmp_path)),
             nprocs=world, join=True)
    for rank in range(world):
        ok, n = np.load(tmp_path / f'p{rank}.npy')
        assert ok == 1 and n == nsteps + 2


def _worker_kmax(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyratbay_amd import dist as pbd
    # per-(layer, row) maxima as the bit patterns of non-negative doubles, one shard's view each
    rng = np.random.default_rng(100 + rank)
    local = 10.0**rng.uniform(-30, -15, 24)
    local[rank::world] = 0.0                      # rows this shard saw no line of
    bits = torch.from_numpy(local.view(np.int64).copy())
    pbd.kmax_allreduce()(bits)
    np.save(os.path.join(tmp, f'k{rank}.npy'), bits.numpy().view(np.float64))
    np.save(os.path.join(tmp, f'l{rank}.npy'), local)
    dist.barrier()
    dist.destroy_process_group()


def test_kmax_exchange_two_ranks(tmp_path):
    """dist.kmax_allreduce: integer MAX over the ranks of the double bit patterns = the maximum
    of the doubles (the exchange step of the two-phase shard extinction)."""
    world = 2
    mp.spawn(_worker_kmax, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    local = [np.load(tmp_path / f'l{r}.npy') for r in range(world)]
    want = np.maximum(local[0], local[1])
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f'k{rank}.npy'), want)


def _bench(*argv, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv),
                          capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_bench_self_launcher_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent (which never touches
    the GPU) starts the two ranks itself with the rendezvous variables of
    torch.distributed.run, relays rank 0's single JSON line and returns 0.  The child body here
    is the launcher's CPU self-test (gloo all-reduce of rank + 1)."""
    import json
    r = _bench('--gpus', '2', '--selftest-launch', 'ok')
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                              # rank 0 only
    rec = json.loads(lines[0])
    assert rec['world'] == 2 and rec['sum'] == 3.0 and rec['local_rank'] == 0
    assert rec['master'].startswith('127.0.0.1:')


def test_bench_self_launcher_propagates_failure():
    """A rank that dies takes the job down with its exit code (the others are terminated, not
    left waiting in a rendezvous), and a launcher-provided WORLD_SIZE that disagrees with
    --gpus is still refused."""
    r = _bench('--gpus', '3', '--selftest-launch', 'fail')
    assert r.returncode == 3
    assert 'rank 1 exited with code 3' in r.stderr
    r = _bench('--gpus', '2', '--selftest-launch', 'ok', env={'WORLD_SIZE': '4', 'RANK': '0'})
    assert r.returncode != 0
