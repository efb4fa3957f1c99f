"""Sharding of one spectrum across the GPUs of one node (one process per GPU).

Two decompositions are provided:

* wavenumber shards: every stage after line broadening is
  column-independent and the line list is replicated, so rank r computes the output
  samples [bounds[r], bounds[r+1]) of the GLOBAL grid with no exchange in the middle of
  the path (`SpectrumGather` closes the step with an all-gather).  Each shard still walks
  all layers, re-derives the per-line records and carries a +-cutoff halo of lines, so
  the fixed part of a step does not shrink with the number of ranks.

* layer-sharded extinction (`LayerShardedTransit`): the extinction of different layers is
  independent -- it is the axis the reference itself parallelises over with fork
  (pyratbay/pyrat/line_by_line.py:232-246).  Rank r computes ec for the layers
  r, r+N, r+2N, ... over the FULL grid (deep and shallow layers interleave, so the load
  balances), one all-to-all turns [my layers, all columns] into [all layers, my columns],
  the column stages (optical depth, spectrum) run on the wavenumber shard, and the
  all-gather re-assembles the spectrum.  Per-rank work is 1/N of the single-GPU work with
  the same tile geometry and no halo; the result is bit-identical to the single-GPU
  spectrum because every ec row and every column is computed by the same arithmetic.

Band fluxes: all-reduce(SUM) of per-shard partial trapezoids (`allreduce_bandflux`).
RCCL over xGMI on GPUs, gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def _via_host(group=None):
    """True when the process group cannot move device tensors (gloo): used only to
    rehearse the multi-rank path on a box with fewer GPUs than ranks."""
    return dist.is_initialized() and dist.get_backend(group) == 'gloo'


def all_gather_flat(recv, send, group=None):
    if _via_host(group) and send.is_cuda:
        r, s_ = recv.cpu(), send.cpu()
        dist.all_gather_into_tensor(r, s_, group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def all_to_all_flat(recv, send, group=None):
    if _via_host(group) and send.is_cuda:
        r, s_ = recv.cpu(), send.cpu()
        dist.all_to_all_single(r, s_, group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group)


def shard_bounds(nwave, world):
    """Contiguous, balanced shards: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(int(nwave), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


class SpectrumGather:
    """Re-assembles the full spectrum from equal-size padded shards with one
    all_gather_into_tensor per step (buffers allocated once)."""

    def __init__(self, nwave, world, rank, device, group=None):
        self.bounds = shard_bounds(nwave, world)
        self.world, self.rank, self.group = world, rank, group
        self.nwave = int(nwave)
        self.pad = int(np.max(np.diff(self.bounds)))
        self.send = torch.zeros(self.pad, dtype=torch.float64, device=device)
        self.recv = torch.zeros(world * self.pad, dtype=torch.float64, device=device)
        self.full = torch.empty(self.nwave, dtype=torch.float64, device=device)

    @property
    def wbegin(self):
        return int(self.bounds[self.rank])

    @property
    def wcount(self):
        return int(self.bounds[self.rank + 1] - self.bounds[self.rank])

    def __call__(self, local):
        """local[wcount] -> full[nwave] on every rank."""
        assert local.shape[0] == self.wcount
        if self.world == 1:
            self.full.copy_(local)
            return self.full
        self.send[:self.wcount].copy_(local)
        all_gather_flat(self.recv, self.send, self.group)
        blocks = self.recv.view(self.world, self.pad)
        base, rem = divmod(self.nwave, self.world)          # see shard_bounds
        if rem:
            self.full[:rem * (base + 1)].view(rem, base + 1).copy_(blocks[:rem, :base + 1])
        if base:
            self.full[rem * (base + 1):].view(self.world - rem, base).copy_(blocks[rem:, :base])
        return self.full


def allreduce_bandflux(partial, heights=None, group=None):
    """Sum the per-shard partial band integrals, then apply the pass-band heights
    (spec_tools.py:232-233) -> bandflux on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial * heights if heights is not None else partial


def layer_exchange(ec_mine, nlayers, nwave, world, rank, group=None, buffers=None):
    """[my layers (r, r+N, ...), all columns] -> [all layers, my columns].

    ec_mine: tensor [Lp, nwave] with Lp = ceil(nlayers/world) rows (rows beyond this
    rank's share are padding).  Returns (ec_cols[nlayers, wcount], bounds).  One
    all_to_all_single of equal blocks; layer l = src + world*j lands in row j*world + src,
    i.e. the natural layer order."""
    lp = -(-nlayers // world)
    bounds = shard_bounds(nwave, world)
    wp = int(np.max(np.diff(bounds)))
    assert ec_mine.shape == (lp, nwave)
    if buffers is None:
        buffers = (torch.zeros((world, lp, wp), dtype=ec_mine.dtype, device=ec_mine.device),
                   torch.empty((world, lp, wp), dtype=ec_mine.dtype, device=ec_mine.device))
    send, recv = buffers
    # balanced shards are `rem` blocks of base+1 columns followed by blocks of `base` columns:
    # two strided copies pack all destinations (not one small kernel per destination)
    base, rem = divmod(int(nwave), int(world))
    if rem:
        send[:rem, :, :base + 1].copy_(
            ec_mine.as_strided((rem, lp, base + 1), (base + 1, nwave, 1), ec_mine.storage_offset()))
    if base:
        send[rem:, :, :base].copy_(
            ec_mine.as_strided((world - rem, lp, base), (base, nwave, 1),
                               ec_mine.storage_offset() + rem * (base + 1)))
    if world > 1:
        all_to_all_flat(recv.view(-1), send.view(-1), group)
    else:
        recv.copy_(send)
    wcount = int(bounds[rank + 1] - bounds[rank])
    # recv[src, j, :] is layer src + world*j
    ec_cols = recv.permute(1, 0, 2).reshape(lp * world, wp)[:nlayers, :wcount]
    return ec_cols.contiguous(), bounds


class LayerShardedTransit:
    """One transit spectrum per step on `world` GPUs: layer-sharded LBL extinction, one
    all-to-all, wavenumber-sharded optical depth + transmission, one all-gather."""

    def __init__(self, case, world, rank, group=None):
        from . import engine
        self.engine = engine
        self.world, self.rank, self.group = world, rank, group
        g, atm, iso = case['grid'], case['atm'], case['iso']
        self.nwave, self.nlayers = g['nwave'], atm['nlayers']
        self.layers = np.arange(rank, self.nlayers, world)
        self.lp = -(-self.nlayers // world)
        self.model = engine.LBLSpectrum(case, rt_path='transit')      # full grid, full plan
        idx = torch.as_tensor(self.layers, device='cuda')
        self.temp = self.model.temp[idx].contiguous()
        self.dens = self.model.dens[idx].contiguous()
        self.isoz = self.model.isoz[:, idx].contiguous()
        self.ec = torch.zeros((self.lp, 1, self.nwave), dtype=torch.float64, device='cuda')
        self.gather = SpectrumGather(self.nwave, world, rank, 'cuda', group)
        wp = self.gather.pad
        self.buffers = (torch.zeros((world, self.lp, wp), dtype=torch.float64, device='cuda'),
                        torch.empty((world, self.lp, wp), dtype=torch.float64, device='cuda'))
        self.lbl = self.model.lbl

    def step(self):
        m, e = self.model, self.engine
        n = len(self.layers)
        if n:
            m.lbl.extinction(self.temp, self.dens, self.isoz, add=True, out=self.ec[:n])
        ec_cols, _ = layer_exchange(self.ec.view(self.lp, self.nwave), self.nlayers,
                                    self.nwave, self.world, self.rank, self.group,
                                    self.buffers)
        spec, self.depth, self.ideep = e.transit_spectrum(
            ec_cols, m.raypath, m.radius, m.rstar, m.itop, self.nlayers, m.maxdepth)
        return self.gather(spec)


def walker_slice(nwalkers, world, rank):
    """Replica parallelism for retrieval: every GPU holds the
    full cross-section table and evaluates a contiguous slice of the walker batch."""
    b = shard_bounds(nwalkers, world)
    return int(b[rank]), int(b[rank + 1])


def gather_walkers(local, nwalkers, world, rank, group=None):
    """local[n_r, nbands] -> bandflux[nwalkers, nbands] on every rank (one all-gather of
    equal padded blocks)."""
    b = shard_bounds(nwalkers, world)
    pad = int(np.max(np.diff(b)))
    nb = local.shape[1]
    send = torch.zeros((pad, nb), dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    if world == 1:
        return local.clone()
    recv = torch.empty((world * pad, nb), dtype=local.dtype, device=local.device)
    all_gather_flat(recv.view(-1), send.view(-1), group)
    blocks = recv.view(world, pad, nb)
    return torch.cat([blocks[r, :int(b[r + 1] - b[r])] for r in range(world)], dim=0)
