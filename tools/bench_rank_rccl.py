#!/usr/bin/env python3
"""What ONE rank of an N-GPU wavenumber-sharded run costs per spectrum on a one-GPU box, the
collectives INCLUDED as far as one GPU can show them: a one-rank RCCL process group, the
all-reduce(MAX) of the maxima and the closing all-gather issued exactly as dist.ShardPipeline issues
them (same tensors, same streams; the all-gather of this rank's padded shard into a buffer of its
own size).  What it measures beyond tools/bench_wshard.py: the host's submission cost of a spectrum
(library calls + torch ops + two collectives) against the 0.18 ms of GPU work of a 1/8 shard.
usage: python tools/bench_rank_rccl.py <world> [workload] [streams] [stack]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch
import torch.distributed as dist

import bench
from pyratbay_amd import dist as pbdist
from pyratbay_amd import engine


class OneRankGather(pbdist.SpectrumGather):
    """SpectrumGather of rank r of `world`, the all-gather on a one-rank group (own block only)."""

    def __call__(self, local):
        if local.data_ptr() != self.send.data_ptr():
            self.send[:self.wcount].copy_(local)
        pbdist.all_gather_flat(self.recv[:self.pad], self.send)
        if self.uniform:
            return self.full
        blocks = self.recv.view(self.world, self.pad)
        base, rem = divmod(self.nwave, self.world)
        if rem:
            self.full[:rem * (base + 1)].view(rem, base + 1).copy_(blocks[:rem, :base + 1])
        if base:
            self.full[rem * (base + 1):].view(self.world - rem, base).copy_(blocks[rem:, :base])
        return self.full


class OneRankStackGather(pbdist.StackGather):
    """StackGather of rank r of `world` with the all-gather on a one-rank group: this rank's K x pad
    block into a buffer of its own size; the sorting copy runs at its real size."""

    def _exchange(self):
        pbdist.all_gather_flat(self.recv[0].view(-1), self.send.view(-1))


def measure(world, name='c2', streams=3, steps=200, voigt=None, lines=None, collectives=(True, False),
            log=None, prime_seconds=0.3, keep=False, stack=1):
    """ms per spectrum (and host submission ms) of rank world // 2 of `world`, `streams` spectra in
    flight, with the collectives through the (already initialised, one-rank) RCCL group and/or
    without them.  Returns {label: (ms_per_spectrum, host_ms_per_spectrum)}."""
    case = bench.make_case(bench.WORKLOADS[name])
    nwave = case['grid']['nwave']
    r = world // 2
    if stack > 1:
        # K atmospheres per extinction call (dist.StackedShard): K x streams spectra in flight
        pipe = pbdist.ShardPipeline(case, world, r, depth=streams, kmax_exchange=True, voigt=voigt,
                                    lines=lines, stack=stack)
        pipe.gathers = [OneRankStackGather(nwave, world, r, stack, 'cuda') for _ in range(streams)]
        for m, g in zip(pipe.models, pipe.gathers):
            m.spectrum_out = g.slots
    else:
        pipe = pbdist.ShardPipeline(case, world, r, depth=streams, kmax_exchange=True, voigt=voigt,
                                    lines=lines,
                                    timestamps=os.environ.get('PB_TIMESTAMPS', '0') == '1')
        pipe.gathers = [OneRankGather(nwave, world, r, 'cuda', uniform=True) for _ in range(streams)]
        for m, g in zip(pipe.models, pipe.gathers):
            m.spectrum_out = g.slot if g.uniform else None
    out = {}
    for coll in collectives:
        label = 'collectives through RCCL (one rank)' if coll else 'no collectives'
        for i, m in enumerate(pipe.models):
            m.kmax_exchange = pbdist.kmax_allreduce() if coll else (lambda t: None)
        if not coll and stack > 1:
            for m in pipe.models:
                m.spectrum_out = [None] * stack
            pipe.gathers = [pbdist.StackGather(nwave, 1, 0, stack, 'cuda') for _ in range(streams)]
            for g, m in zip(pipe.gathers, pipe.models):   # (world 1: returns the local shards)
                g.bounds = np.array([0, m.wcount])
        elif not coll:
            for m in pipe.models:
                m.spectrum_out = None
            pipe.gathers = [pbdist.SpectrumGather(nwave, 1, 0, 'cuda') for _ in range(streams)]
            for g, m in zip(pipe.gathers, pipe.models):   # (world 1: returns the local shard)
                g.bounds = np.array([0, m.wcount])
        # warm-up: every context's workspaces exist and the chip is at its working clocks (after an
        # idle period the first few hundred spectra run at lower clocks)
        tw = time.perf_counter()
        while True:
            for _ in range(4 * streams):
                pipe.submit()
            pipe.flush()
            torch.cuda.synchronize()
            if time.perf_counter() - tw >= prime_seconds:
                break
        nsub = -(-steps // stack)
        t0 = time.perf_counter()
        for _ in range(nsub):
            pipe.submit()
        t_submit = time.perf_counter() - t0
        pipe.flush()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        done = nsub * stack
        out[label] = (1e3 * t_all / done, 1e3 * t_submit / done)
        if log:
            log(f'{name} rank {r}/{world}, {streams} x {stack} in flight, {label}: '
                f'{1e3 * t_all / done:.4f} ms/spectrum, host submission '
                f'{1e3 * t_submit / done:.3f} ms/spectrum')
    if keep:
        out['voigt'] = pipe.models[0].voigt
    return out


def main():
    world = int(sys.argv[1])
    name = sys.argv[2] if len(sys.argv) > 2 else 'c2'
    streams = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    steps = int(os.environ.get('PB_WSHARD_STEPS', '200'))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    stack = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    measure(world, name, streams, steps, log=lambda s: print(s, flush=True), stack=stack)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
