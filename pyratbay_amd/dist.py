"""Wavenumber sharding across the GPUs of one node (one process per GPU).

Every stage after line broadening is column-independent, and the
line list is replicated, so rank r computes the output samples [bounds[r], bounds[r+1])
of the GLOBAL grid with no exchange in the middle of the path.  Two collectives close a
step: an all-gather of the spectrum shards (RCCL over xGMI on GPUs, gloo in the CPU tests)
and an all-reduce(SUM) of the per-band partial trapezoids.
"""
import numpy as np
import torch
import torch.distributed as dist


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
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        blocks = self.recv.view(self.world, self.pad)
        for r in range(self.world):
            n = int(self.bounds[r + 1] - self.bounds[r])
            self.full[self.bounds[r]:self.bounds[r + 1]].copy_(blocks[r, :n])
        return self.full


def allreduce_bandflux(partial, heights=None, group=None):
    """Sum the per-shard partial band integrals, then apply the pass-band heights
    (spec_tools.py:232-233) -> bandflux on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial * heights if heights is not None else partial
