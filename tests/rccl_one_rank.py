"""Child process of tests/test_gpu_rccl.py: a ONE-rank RCCL process group on cuda:0.

A one-GPU box cannot show a collective moving data between ranks, but it can show that every
collective of pyratbay_amd/dist.py is accepted by torch's ProcessGroupNCCL (= RCCL on ROCm)
on the very tensors the multi-GPU run hands it -- among them the int64 tensor that aliases
library memory through __cuda_array_interface__ (not an allocation of torch's caching
allocator) -- issued from side streams like dist.ShardPipeline does, and that the streams are
ordered (results equal the collective-free run).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from pyratbay_amd import engine, synth
    from pyratbay_amd import dist as pbdist

    engine.require_gpu()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    out = {'backend': dist.get_backend()}
    case = synth.lbl_case(3001, 14, 12000, wnosamp=24, nlor=20, ndop=10, extent=80.0,
                          cutoff=4.0, niso=2, seed=11)
    plain = engine.LBLSpectrum(case, rt_path='transit')
    want = plain.run().clone()
    want_ec = plain.ec.clone()

    # (1) the two-phase shard extinction with the all-reduce(MAX) of the per-row maxima on the
    #     aliasing tensor, three spectra in flight on side streams
    models = [engine.LBLSpectrum(case, rt_path='transit', voigt=plain.voigt, lines=plain.lines)
              for _ in range(3)]
    streams = engine.side_streams(3)
    for m in models:
        m.kmax_exchange = pbdist.kmax_allreduce()
    got = []
    for i in range(6):
        m, s = models[i % 3], streams[i % 3]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            got.append(m.run().clone())
    torch.cuda.synchronize()
    kmax = models[0].lbl.kmax_tensor()
    assert kmax.dtype == torch.int64 and kmax.is_cuda
    out['kmax_words'] = int(kmax.numel())
    out['kmax_positive'] = int((kmax > 0).sum().item())
    assert out['kmax_positive'] > 0
    err = max(float(torch.max(torch.abs(g / want - 1)).item()) for g in got)
    out['two_phase_vs_one_call'] = err
    assert err <= 1e-12, err

    # (2) closing all-gather and band all-reduce, as SpectrumGather / allreduce_bandflux issue them
    send = want[:1000].clone()
    recv = torch.zeros(1000, dtype=torch.float64, device='cuda')
    pbdist.all_gather_flat(recv, send)
    work = pbdist.all_gather_flat_async(recv, send)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    part = torch.arange(4, dtype=torch.float64, device='cuda')
    flux = pbdist.allreduce_bandflux(part.clone(), heights=2 * torch.ones_like(part))
    assert torch.equal(flux, 2 * part)

    # (3) the layer decomposition's exchange pipeline: asynchronous all-to-all + all-gather on
    #     the process group's stream, three stages in flight
    sharded = pbdist.LayerShardedTransit(case, 1, 0, voigt=plain.voigt, lines=plain.lines)
    outs = []
    for _ in range(5):
        r = sharded.submit()
        if r is not None:
            outs.append(r.clone())
    outs += [o.clone() for o in sharded.flush()]
    torch.cuda.synchronize()
    assert len(outs) == 5
    err = max(float(torch.max(torch.abs(o / want - 1)).item()) for o in outs)
    out['layer_pipeline_vs_single'] = err
    assert err <= 1e-12, err
    assert torch.equal(sharded.step(), outs[-1])

    # (5) dist.ShardPipeline as rank 1 of 3 (equal-size shards written straight into the gather
    #     buffer, the all-reduce of the maxima, each spectrum's all-gather issued behind the next
    #     spectrum's all-reduce), the all-gather on this one-rank group moving the rank's own block
    class OneRankGather(pbdist.SpectrumGather):
        def __call__(self, local):
            assert local.data_ptr() == self.send.data_ptr()      # written in place by the model
            pbdist.all_gather_flat(self.recv[self.rank * self.pad:(self.rank + 1) * self.pad],
                                   self.send)
            return self.full

    nwave = case['grid']['nwave']
    pipe = pbdist.ShardPipeline(case, 3, 1, depth=3, kmax_exchange=True, voigt=plain.voigt,
                                lines=plain.lines, timestamps=False)
    pipe.gathers = [OneRankGather(nwave, 3, 1, 'cuda', uniform=True) for _ in range(3)]
    for m, g in zip(pipe.models, pipe.gathers):
        assert g.uniform
        m.spectrum_out = g.slot
    g0 = pipe.gathers[0]
    fulls = []
    for i in range(7):
        r = pipe.submit()
        assert (r is None) == (i == 0)
        if r is not None:
            fulls.append(r[0].clone())
    fulls.append(pipe.flush()[0].clone())
    torch.cuda.synchronize()
    assert len(fulls) == 7
    a, b = g0.wbegin, g0.wbegin + g0.wcount
    err = max(float(torch.max(torch.abs(f[a:b] / want[a:b] - 1)).item()) for f in fulls)
    out['shard_pipeline_vs_single'] = err
    assert err <= 1e-12, err

    # (6) the stacked form as rank 1 of 3: K = 2 atmospheres per submission, ONE all-reduce of their
    #     maxima and ONE all-gather of their two shards (dist.StackGather; on this one-rank group the
    #     all-gather moves the rank's own K x pad block), the sorting copy at its real size
    class OneRankStackGather(pbdist.StackGather):
        def _exchange(self):
            pbdist.all_gather_flat(self.recv[self.rank].view(-1), self.send.view(-1))

    spipe = pbdist.ShardPipeline(case, 3, 1, depth=2, kmax_exchange=True, voigt=plain.voigt,
                                 lines=plain.lines, stack=2)
    spipe.gathers = [OneRankStackGather(nwave, 3, 1, 2, 'cuda') for _ in range(2)]
    for m, g in zip(spipe.models, spipe.gathers):
        assert g.uniform
        m.spectrum_out = g.slots
    stacked = []
    for i in range(4):
        r = spipe.submit()
        assert (r is None) == (i == 0)
        if r is not None:
            stacked += [x.clone() for x in r[0]]
    stacked += [x.clone() for x in spipe.flush()[0]]
    torch.cuda.synchronize()
    assert len(stacked) == 8
    err = max(float(torch.max(torch.abs(f[a:b] / want[a:b] - 1)).item()) for f in stacked)
    out['stacked_pipeline_vs_single'] = err
    assert err <= 1e-12, err
    # (the blocks of the other ranks were never received: they stay zero)
    assert all(float(f[:a].abs().max()) == 0.0 for f in stacked)

    # (4) walkers gathered over the ranks (replica form of the retrieval batch)
    local = torch.rand((7, 3), dtype=torch.float64, device='cuda')
    assert torch.equal(pbdist.gather_walkers(local, 7, 1, 0), local)
    np.testing.assert_allclose(plain.ec.cpu().numpy(), want_ec.cpu().numpy(), rtol=0)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
