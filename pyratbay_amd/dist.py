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
  the same tile geometry and no halo.  Every rank gets the same spectrum bit for bit; against
  the single-GPU spectrum it agrees to ~1e-13 (a rank-size launch splits the phases of a tile
  between more workgroups, which changes the association of the per-sample sums -- bit for bit
  only when both runs pick the same kernel and split, e.g. PB_STAGE_SPLIT=1).

Band fluxes: all-reduce(SUM) of per-shard partial trapezoids (`allreduce_bandflux`).
RCCL over xGMI on GPUs, gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def _range(name):
    """rocTX range around a collective (engine.profiler_range; the import is deferred so that the
    gloo tests of this module run without the HIP library)."""
    from .engine import profiler_range
    return profiler_range(name)


class _NoRange:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _coll_range(name, tensor):
    return _range(name) if tensor.is_cuda else _NoRange()


def _via_host(group=None):
    """True when the process group cannot move device tensors (gloo): used only to
    rehearse the multi-rank path on a box with fewer GPUs than ranks."""
    return dist.is_initialized() and dist.get_backend(group) == 'gloo'


def all_gather_flat(recv, send, group=None):
    with _coll_range('all_gather', send):
        if _via_host(group) and send.is_cuda:
            r, s_ = recv.cpu(), send.cpu()
            dist.all_gather_into_tensor(r, s_, group=group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=group)


def all_to_all_flat(recv, send, group=None):
    with _coll_range('all_to_all', send):
        if _via_host(group) and send.is_cuda:
            r, s_ = recv.cpu(), send.cpu()
            dist.all_to_all_single(r, s_, group=group)
            recv.copy_(r)
        else:
            dist.all_to_all_single(recv, send, group=group)


class _Done:
    """Stand-in for a collective that has already completed (host-staged rehearsal)."""

    def wait(self):
        return True


def all_gather_flat_async(recv, send, group=None):
    """Start the all-gather and return its Work: the collective runs on the process group's
    own stream, the caller's stream goes on and meets it again at `work.wait()`."""
    if not dist.is_initialized():
        recv.copy_(send.repeat(recv.numel() // send.numel()))
        return _Done()
    if _via_host(group) and send.is_cuda:
        all_gather_flat(recv, send, group)
        return _Done()
    with _coll_range('all_gather_start', send):
        return dist.all_gather_into_tensor(recv, send, group=group, async_op=True)


def all_to_all_flat_async(recv, send, group=None):
    if not dist.is_initialized():                  # single process: the exchange is a copy
        recv.copy_(send)
        return _Done()
    if _via_host(group) and send.is_cuda:
        all_to_all_flat(recv, send, group)
        return _Done()
    with _coll_range('all_to_all_start', send):
        return dist.all_to_all_single(recv, send, group=group, async_op=True)


def shard_bounds(nwave, world):
    """Contiguous, balanced shards: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(int(nwave), int(world))
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def uniform_bounds(nwave, world):
    """Shards of ONE size, pad = ceil(nwave / world), the last one shorter: rank r owns
    [r pad, min((r + 1) pad, nwave)).  The gathered buffer [world x pad] then IS the spectrum
    (followed by the last block's padding): no unpacking.  None when a rank would own nothing."""
    pad = -(-int(nwave) // int(world))
    if (world - 1) * pad >= nwave:
        return None
    return np.minimum(np.arange(world + 1, dtype=np.int64) * pad, int(nwave))


class SpectrumGather:
    """Re-assembles the full spectrum from equal-size padded shards with one
    all_gather_into_tensor per step (buffers allocated once).

    uniform=True (the wavenumber decomposition's pipeline): shards of one size (`uniform_bounds`),
    the model writes its shard straight into `slot` (= this rank's part of the send buffer,
    LBLSpectrum.spectrum_out) and the receive buffer is the result -- ONE collective and no copy
    per spectrum; the balanced form (shards that differ by at most one sample) copies the shard in
    and unpacks two strided blocks, three small launches and ~50 us of host time per spectrum,
    which at 8 ranks is a quarter of what a rank spends on a C2 spectrum."""

    def __init__(self, nwave, world, rank, device, group=None, uniform=False):
        self.uniform = bool(uniform) and world > 1 and uniform_bounds(nwave, world) is not None
        self.bounds = uniform_bounds(nwave, world) if self.uniform else shard_bounds(nwave, world)
        self.world, self.rank, self.group = world, rank, group
        self.nwave = int(nwave)
        self.pad = int(np.max(np.diff(self.bounds)))
        self.send = torch.zeros(self.pad, dtype=torch.float64, device=device)
        self.recv = torch.zeros(world * self.pad, dtype=torch.float64, device=device)
        self.full = (self.recv[:self.nwave] if self.uniform else
                     torch.empty(self.nwave, dtype=torch.float64, device=device))

    @property
    def wbegin(self):
        return int(self.bounds[self.rank])

    @property
    def wcount(self):
        return int(self.bounds[self.rank + 1] - self.bounds[self.rank])

    @property
    def slot(self):
        """Where a model may write this rank's shard directly (uniform form)."""
        return self.send[:self.wcount]

    def __call__(self, local):
        """local[wcount] -> full[nwave] on every rank."""
        assert local.shape[0] == self.wcount
        if self.world == 1:
            return local                          # nothing to assemble
        if local.data_ptr() != self.send.data_ptr():
            self.send[:self.wcount].copy_(local)
        all_gather_flat(self.recv, self.send, self.group)
        if self.uniform:
            return self.full                      # the receive buffer is the spectrum
        blocks = self.recv.view(self.world, self.pad)
        base, rem = divmod(self.nwave, self.world)          # see shard_bounds
        if rem:
            self.full[:rem * (base + 1)].view(rem, base + 1).copy_(blocks[:rem, :base + 1])
        if base:
            self.full[rem * (base + 1):].view(self.world - rem, base).copy_(blocks[rem:, :base])
        return self.full


class StackGather:
    """The closing collective of a StackedShard: the K shards of a stacked submission leave in ONE
    all_gather_into_tensor (K x pad doubles per rank) instead of K, and one strided copy sorts the
    blocks [world][K][pad] into K contiguous spectra [K][world x pad].

    Why: the K all-gathers of ~100 kB each are latency-bound launches on the communicator's one
    stream -- K + 1 collectives per stacked submission of a rank that spends ~0.44 ms on it at
    C2 / 8 ranks; this form issues two.  Same padding rule as SpectrumGather(uniform=True): shards
    of one size, the last one shorter (balanced shards, unpacked rank by rank, when a rank would
    own nothing otherwise); `slots[k]` (where atmosphere k's transit call writes its shard) is part
    of the send buffer and nothing is packed on the way in."""

    def __init__(self, nwave, world, rank, stack, device, group=None):
        self.bounds = uniform_bounds(nwave, world)
        self.uniform = self.bounds is not None
        if not self.uniform:                      # (a rank would own nothing: balanced shards,
            self.bounds = shard_bounds(nwave, world)   # unpacked rank by rank)
        self.world, self.rank, self.group, self.stack = world, rank, group, int(stack)
        self.nwave = int(nwave)
        self.pad = int(np.max(np.diff(self.bounds)))
        K = self.stack
        self.send = torch.zeros((K, self.pad), dtype=torch.float64, device=device)
        self.recv = torch.zeros((world, K, self.pad), dtype=torch.float64, device=device)
        self.sorted = torch.zeros((K, world * self.pad), dtype=torch.float64, device=device)
        self.full = [self.sorted[k, :self.nwave] for k in range(K)]

    @property
    def wbegin(self):
        return int(self.bounds[self.rank])

    @property
    def wcount(self):
        return int(self.bounds[self.rank + 1] - self.bounds[self.rank])

    @property
    def slots(self):
        """Where the model writes the K shards of this rank (views of the send buffer)."""
        return [self.send[k, :self.wcount] for k in range(self.stack)]

    def _exchange(self):
        all_gather_flat(self.recv.view(-1), self.send.view(-1), self.group)

    def __call__(self, local):
        """local: K tensors [wcount] -> K full spectra [nwave] on every rank."""
        assert len(local) == self.stack
        if self.world == 1:
            return list(local)
        for k, x in enumerate(local):
            assert x.shape[0] == self.wcount
            if x.data_ptr() != self.send[k].data_ptr():
                self.send[k, :self.wcount].copy_(x)
        self._exchange()
        if self.uniform:
            self.sorted.view(self.stack, self.world, self.pad).copy_(self.recv.transpose(0, 1))
        else:
            for r in range(self.world):
                a, b = int(self.bounds[r]), int(self.bounds[r + 1])
                self.sorted[:, a:b].copy_(self.recv[r, :, :b - a])
        return self.full


def kmax_allreduce(group=None):
    """The exchange step of the two-phase shard extinction (engine.LBL.extinction_begin/_end):
    returns a function that all-reduces (MAX) the int64 view of the per-row maxima in place --
    nlayers x rows words, the only mid-path collective of the wavenumber decomposition
    (SURVEY 8e option 1; _extcoeff.c:225,265)."""
    def exchange(kmax):
        with _coll_range('all_reduce_kmax', kmax):
            if _via_host(group):
                host = kmax.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
                kmax.copy_(host)
            else:
                dist.all_reduce(kmax, op=dist.ReduceOp.MAX, group=group)
    return exchange


def allreduce_bandflux(partial, heights=None, group=None):
    """Sum the per-shard partial band integrals, then apply the pass-band heights
    (spec_tools.py:232-233) -> bandflux on every rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        with _coll_range('all_reduce_bandflux', partial):
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

    def __init__(self, case, world, rank, group=None, voigt=None, lines=None):
        from . import engine
        self.engine = engine
        self.world, self.rank, self.group = world, rank, group
        g, atm, iso = case['grid'], case['atm'], case['iso']
        self.nwave, self.nlayers = g['nwave'], atm['nlayers']
        self.layers = np.arange(rank, self.nlayers, world)
        self.lp = -(-self.nlayers // world)
        self.model = engine.LBLSpectrum(case, rt_path='transit', voigt=voigt,
                                        lines=lines)                  # full grid, full plan
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
        self._idx = idx

    def set_atmosphere(self, temp, dens, isoz, radius=None):
        """New atmosphere for the next step()/submit(): forwards to the model (radius, ray
        paths, full-grid state) and refreshes this rank's layer slices IN PLACE, so the
        pipeline's buffers and any captured pointers stay valid."""
        self.model.set_atmosphere(temp, dens, isoz, radius)
        self.temp.copy_(self.model.temp[self._idx])
        self.dens.copy_(self.model.dens[self._idx])
        self.isoz.copy_(self.model.isoz[:, self._idx])

    def step(self):
        m, e = self.model, self.engine
        n = len(self.layers)
        if n:
            if m.resolution:
                self.ec[:n].zero_()      # that mode ACCUMULATES into ec (like the reference)
            m.lbl.extinction(self.temp, self.dens, self.isoz, add=True, out=self.ec[:n])
        ec_cols, _ = layer_exchange(self.ec.view(self.lp, self.nwave), self.nlayers,
                                    self.nwave, self.world, self.rank, self.group,
                                    self.buffers)
        spec, self.depth, self.ideep = e.transit_spectrum(
            ec_cols, m.raypath, m.radius, m.rstar, m.itop, self.nlayers, m.maxdepth)
        return self.gather(spec)

    # -- the same step, software-pipelined over consecutive spectra (ExchangePipeline) ------
    def _produce(self, ec_mine):
        n = len(self.layers)
        if n:
            if self.model.resolution:
                self.ec[:n].zero_()
            self.model.lbl.extinction(self.temp, self.dens, self.isoz, add=True, out=self.ec[:n])

    def _consume(self, ec_cols):
        m = self.model
        spec, self.depth, self.ideep = self.engine.transit_spectrum(
            ec_cols, m.raypath, m.radius, m.rstar, m.itop, self.nlayers, m.maxdepth)
        return spec

    def _pipeline(self):
        if getattr(self, '_pipe', None) is None:
            self._pipe = ExchangePipeline(self.ec.view(self.lp, self.nwave), self.nlayers,
                                          self.world, self.rank, self._produce, self._consume,
                                          self.group)
        return self._pipe

    def submit(self):
        """Enqueue one more spectrum; returns the one submitted two calls ago (or None)."""
        return self._pipeline().submit()

    def flush(self):
        """Complete the spectra still in flight and return them in order."""
        return self._pipeline().flush() if getattr(self, '_pipe', None) is not None else []


class ExchangePipeline:
    """produce -> all-to-all -> consume -> all-gather, software-pipelined over consecutive
    spectra.

    `produce(ec_mine)` fills ec_mine[lp, nwave] (this rank's layers r, r+N, ... over all
    columns) in place; `consume(ec_cols)` maps ec_cols[nlayers, wcount] (all layers, this
    rank's columns) to this rank's part of the spectrum, spec[wcount].  submit(i) enqueues,
    in this order on the caller's stream:
      A(i)   produce, pack, START all-to-all i                    (the collective's stream)
      B(i-1) wait all-to-all i-1, consume, START all-gather i-1
      C(i-2) wait all-gather i-2, unpack -> full spectrum i-2 (returned)
    so both collectives of a spectrum run beside the production of the next one.  Every
    rank issues the collectives in the same order (a2a i, ag i-1, a2a i+1, ...).  Buffers are
    double: a collective works on set i % 2 while set (i+1) % 2 is packed or consumed; set
    i % 2 is next touched by A(i+2) / B(i+2), enqueued after B(i) / C(i) on the caller's
    stream, which the collective's stream waits for when the collective is issued."""

    def __init__(self, ec_mine, nlayers, world, rank, produce, consume, group=None):
        self.ec, self.nlayers, self.world, self.rank = ec_mine, int(nlayers), world, rank
        self.produce, self.consume, self.group = produce, consume, group
        self.lp, self.nwave = ec_mine.shape
        assert self.lp == -(-self.nlayers // world)
        self.bounds = shard_bounds(self.nwave, world)
        self.wcount = int(self.bounds[rank + 1] - self.bounds[rank])
        wp = self.pad = int(np.max(np.diff(self.bounds)))
        kw = dict(dtype=ec_mine.dtype, device=ec_mine.device)
        self.send = [torch.zeros((world, self.lp, wp), **kw) for _ in range(2)]
        self.recv = [torch.zeros((world, self.lp, wp), **kw) for _ in range(2)]
        self.gsend = [torch.zeros(wp, **kw) for _ in range(2)]
        self.grecv = [torch.zeros(world * wp, **kw) for _ in range(2)]
        self.full = [torch.empty(self.nwave, **kw) for _ in range(2)]
        self.a2a, self.ag = [None, None], [None, None]
        self.count = 0

    def _stage_a(self, i):
        self.produce(self.ec)
        send, ec = self.send[i % 2], self.ec
        base, rem = divmod(self.nwave, self.world)
        if rem:
            send[:rem, :, :base + 1].copy_(
                ec.as_strided((rem, self.lp, base + 1), (base + 1, self.nwave, 1),
                              ec.storage_offset()))
        if base:
            send[rem:, :, :base].copy_(
                ec.as_strided((self.world - rem, self.lp, base), (base, self.nwave, 1),
                              ec.storage_offset() + rem * (base + 1)))
        self.a2a[i % 2] = all_to_all_flat_async(self.recv[i % 2].view(-1), send.view(-1),
                                                self.group)

    def _stage_b(self, i):
        self.a2a[i % 2].wait()
        # recv[src, j, :] is layer src + world*j
        ec_cols = self.recv[i % 2].permute(1, 0, 2).reshape(self.lp * self.world, self.pad)
        spec = self.consume(ec_cols[:self.nlayers, :self.wcount].contiguous())
        self.gsend[i % 2][:self.wcount].copy_(spec)
        self.ag[i % 2] = all_gather_flat_async(self.grecv[i % 2], self.gsend[i % 2], self.group)

    def _stage_c(self, i):
        self.ag[i % 2].wait()
        full = self.full[i % 2]
        blocks = self.grecv[i % 2].view(self.world, self.pad)
        base, rem = divmod(self.nwave, self.world)
        if rem:
            full[:rem * (base + 1)].view(rem, base + 1).copy_(blocks[:rem, :base + 1])
        if base:
            full[rem * (base + 1):].view(self.world - rem, base).copy_(blocks[rem:, :base])
        return full

    def submit(self):
        i = self.count
        self.count = i + 1
        self._stage_a(i)
        if i >= 1:
            self._stage_b(i - 1)
        return self._stage_c(i - 2) if i >= 2 else None

    def flush(self):
        out = []
        i = self.count                  # A done for 0..i-1, B for 0..i-2, C for 0..i-3
        if i >= 1:
            self._stage_b(i - 1)
        if i >= 2:
            out.append(self._stage_c(i - 2).clone())
        if i >= 1:
            out.append(self._stage_c(i - 1))
        self.count = 0
        return out


class StackedShard:
    """K atmospheres of ONE wavenumber shard per extinction call: their layers are stacked as
    K x L layers of one plan (layer state, records, gather and combine launched once for all of
    them, ONE all-reduce of the K x L x rows maxima), then K transit calls, each writing its shard
    into a slot of the one send buffer (StackGather).

    Why: a 1/8 shard of C2 is a 0.17-ms gather launch over 7 tiles x 80
    layers plus ~0.09 ms of small launches whose size does not shrink with the number of ranks;
    with K atmospheres per call the gather launch is K times larger (a shorter tail per spectrum)
    and the fixed launches are paid once per K spectra: 0.1543 -> 0.1486 / 0.1466 / 0.1454 ms per
    spectrum for K = 2 / 3 / 4 (tools/stack_probe.py).  The callers this serves have the
    atmospheres at hand: the walkers of a retrieval, the temperature loop of an opacity table.
    Each atmosphere's spectrum equals LBLSpectrum.run()'s of the same atmosphere to rounding
    (~1e-16: a larger launch may split the phases of a tile differently), bit for bit with the
    split pinned (PB_STAGE_SPLIT)"""

    def __init__(self, case, stack, wbegin=0, wcount=None, itop=0, voigt=None, lines=None):
        from . import engine
        self.engine = engine
        g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
        assert g.get('resolution') is None and not g.get('interpolate'), \
            'StackedShard: constant-step grids (the interpolating modes plan per layer)'
        self.stack = int(stack)
        self.nwave, self.nlayers, self.itop = g['nwave'], atm['nlayers'], itop
        self.wbegin = wbegin
        self.wcount = self.nwave - wbegin if wcount is None else wcount
        self.maxdepth, self.rstar = case['maxdepth'], float(atm['rstar'])
        self.voigt = voigt or engine.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'],
                                                      g['ownstep'], g['wnosamp'])
        self.lines = lines or engine.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'],
                                              len(iso['isomass']), g['own'])
        K, L = self.stack, self.nlayers
        self.lbl = engine.LBL(self.voigt, self.lines, g['wn'], g['divisors'], atm['mol_radius'],
                              atm['mol_mass'], iso['isoimol'], iso['isomass'], iso['isoratio'],
                              iso['isoiext'], vg['cutoff'], case['ethresh'], max_layers=K * L)
        dev = engine.dev
        self.temp = dev(np.concatenate([atm['temp']] * K))
        self.dens = dev(np.concatenate([atm['dens']] * K))
        self.isoz = dev(np.concatenate([iso['isoz']] * K, axis=1))
        self.radius = [dev(atm['radius']) for _ in range(K)]
        path = engine.pack_raypath(engine.transit_path(atm['radius'], itop), itop)
        self.raypath = [dev(path) for _ in range(K)]
        self.ec = torch.empty((K * L, 1, self.wcount), dtype=torch.float64, device='cuda')
        self.spectrum_out = [None] * K      # gather slots ([wcount] tensors) or None
        self.kmax_exchange = None           # dist.kmax_allreduce(): the two-phase shard form
        self.spectra = [None] * K

    def set_atmosphere(self, k, temp, dens, isoz, radius=None):
        """Atmosphere k of the stack for the next run() (arguments as LBLSpectrum.set_atmosphere)."""
        e, L = self.engine, self.nlayers
        self.temp[k * L:(k + 1) * L].copy_(e.dev(temp))
        self.dens[k * L:(k + 1) * L].copy_(e.dev(dens))
        self.isoz[:, k * L:(k + 1) * L].copy_(e.dev(isoz))
        if radius is not None:
            self.radius[k].copy_(e.dev(radius))
            self.raypath[k].copy_(e.dev(e.pack_raypath(e.transit_path(radius, self.itop),
                                                       self.itop)))

    def run(self):
        """-> the K spectra (device tensors [wcount]; the gather slots when given)."""
        e, K, L = self.engine, self.stack, self.nlayers
        if self.kmax_exchange is not None:
            self.lbl.extinction_begin(self.temp, self.dens, self.isoz, add=True, out=self.ec,
                                      wbegin=self.wbegin, wcount=self.wcount)
            self.kmax_exchange(self.lbl.kmax_tensor())
            self.lbl.extinction_end()
        else:
            self.lbl.extinction(self.temp, self.dens, self.isoz, add=True, out=self.ec,
                                wbegin=self.wbegin, wcount=self.wcount)
        for k in range(K):
            self.spectra[k], _, _ = e.transit_spectrum(
                self.ec[k * L:(k + 1) * L].view(L, self.wcount), self.raypath[k], self.radius[k],
                self.rstar, self.itop, L, self.maxdepth, out=self.spectrum_out[k])
        return self.spectra


class ShardPipeline:
    """Consecutive, independent spectra of a WAVENUMBER-sharded run kept in flight per rank:
    spectrum i+1's extinction runs on a second HIP stream while spectrum i waits for its
    collectives (the all-reduce of the per-row maxima in the middle of the path, the closing
    all-gather) and finishes its small launches.

    Why: a 1/8 shard of C2 is ~0.27 ms of launches of which the gather -- the only one that
    fills the chip -- is 0.17; every collective is a latency a single spectrum sits through.
    With three spectra in flight a rank delivers one per ~0.165 ms (tools/bench_rank_rccl.py),
    and the N = 1 form of bench.py pipelines in the same way (engine.SpectrumPipeline), so the
    N-GPU and the single-GPU numbers are like for like.

    Every context has its own plan and its own SpectrumGather buffers (equal-size shards, the
    model writes its shard straight into the send buffer, the receive buffer of the one all-gather
    is the spectrum); the Voigt table and the line list are shared.  Every rank issues the
    collectives in the same order on one communicator -- all-reduce(i+1) BEFORE all-gather(i), see
    submit() -- so they pair up across the ranks.  submit() -> (full spectrum on every rank,
    event) of the PREVIOUS submission or None; flush() -> those of the last one, and joins the
    caller's stream.

    stack = K > 1 (transit geometry, constant-step grids): every context is a StackedShard -- K
    atmospheres per extinction call, ONE all-reduce of their maxima, ONE all-gather of their K
    shards (StackGather); submit()
    then enqueues K spectra and returns ([K full spectra], event) of the previous submission."""

    def __init__(self, case, world, rank, depth=2, group=None, kmax_exchange=True,
                 voigt=None, lines=None, rt_path='transit', stack=1, **model_kw):
        from . import engine
        nwave = case['grid']['nwave']
        self.world, self.rank, self.group = world, rank, group
        self.stack = int(stack)
        if self.stack > 1:
            assert rt_path == 'transit', 'stack > 1: transit geometry'
            self.gathers = [StackGather(nwave, world, rank, self.stack, 'cuda', group)
                            for _ in range(depth)]
            g0 = self.gathers[0]
            first = StackedShard(case, self.stack, g0.wbegin, g0.wcount,
                                 itop=model_kw.get('itop', 0), voigt=voigt, lines=lines)
            self.models = [first] + [StackedShard(case, self.stack, g0.wbegin, g0.wcount,
                                                  itop=model_kw.get('itop', 0),
                                                  voigt=first.voigt, lines=first.lines)
                                     for _ in range(depth - 1)]
            for m, g in zip(self.models, self.gathers):
                m.lbl.set_concurrency(depth)
                if kmax_exchange and world > 1:
                    m.kmax_exchange = kmax_allreduce(group)
                if world > 1:
                    m.spectrum_out = g.slots
            self.streams = engine.side_streams(depth)
            self.done = [None] * depth
            self.count = 0
            self.pending = None
            return
        self.gathers = [SpectrumGather(nwave, world, rank, 'cuda', group, uniform=True)
                        for _ in range(depth)]
        g0 = self.gathers[0]
        kw = dict(rt_path=rt_path, wbegin=g0.wbegin, wcount=g0.wcount, **model_kw)
        first = engine.LBLSpectrum(case, voigt=voigt, lines=lines, **kw)
        self.models = [first] + [engine.LBLSpectrum(case, voigt=first.voigt, lines=first.lines,
                                                    **kw) for _ in range(depth - 1)]
        for m, g in zip(self.models, self.gathers):
            m.lbl.set_concurrency(depth)
            if kmax_exchange and world > 1:
                m.kmax_exchange = kmax_allreduce(group)
            if world > 1 and rt_path == 'transit' and getattr(m, 'materialize_depth', True):
                m.spectrum_out = g.slot      # the shard goes straight into the gather buffer
        self.streams = engine.side_streams(depth)
        self.done = [None] * depth
        self.count = 0
        self.pending = None          # (context, shard) whose all-gather has not been issued yet

    def set_atmosphere(self, *args, **kw):
        """One atmosphere for every context (stack > 1: for every slot of every stack; use
        models[j].set_atmosphere(k, ...) for distinct ones)."""
        for m in self.models:
            if self.stack > 1:
                for k in range(self.stack):
                    m.set_atmosphere(k, *args, **kw)
            else:
                m.set_atmosphere(*args, **kw)

    def _close(self):
        """Issue the all-gather(s) of the pending submission on its own stream -> (full, event);
        stack > 1: ([K full spectra], event)."""
        j, local = self.pending
        self.pending = None
        stream = self.streams[j]
        with torch.cuda.stream(stream):
            full = self.gathers[j](local)        # (stack > 1: K shards, one all-gather)
            event = torch.cuda.Event()
            event.record(stream)
        self.done[j] = event
        return full, event

    def submit(self):
        """Enqueue one more spectrum; returns (full spectrum, event) of the spectrum submitted
        BEFORE this one (None for the first): its closing all-gather is issued only now, AFTER this
        spectrum's all-reduce of the maxima.  The collectives of one communicator run on one
        stream in the order they are issued, and each waits there for the kernels that produce
        its input: with the all-gather of spectrum i issued before the all-reduce of spectrum
        i+1, that all-reduce -- and with it the whole gather of spectrum i+1 -- waited for the
        last kernel of spectrum i, and the 'pipeline' ran one spectrum at a time plus its
        collectives.  Every rank issues all-reduce(i+1), all-gather(i), all-reduce(i+2), ... in the
        same order."""
        j = self.count % len(self.models)
        self.count += 1
        model, stream = self.models[j], self.streams[j]
        caller = torch.cuda.current_stream()
        if not caller.query():                   # (an idle stream has nothing to wait for)
            stream.wait_stream(caller)
        with torch.cuda.stream(stream):
            local = model.run()                  # ... all-reduce of the maxima inside
        prev = self._close() if self.pending is not None else None
        self.pending = (j, local)
        return prev

    def flush(self):
        """Close the spectrum still pending and join the caller's stream with every spectrum
        submitted so far; returns (full spectrum, event) of the last one (None if none)."""
        last = self._close() if self.pending is not None else None
        cur = torch.cuda.current_stream()
        for event in self.done:
            if event is not None:
                cur.wait_event(event)
        return last


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
