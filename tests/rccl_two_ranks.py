"""Child process of tests/test_gpu_rccl.py::test_two_ranks_over_rccl (needs two GPUs): rank r of
2 on cuda:r over RCCL.  The wavenumber decomposition as bench.py runs it (dist.ShardPipeline: two-
phase shards, the all-reduce(MAX) of the maxima on the aliasing tensor, equal-size shards gathered
by one all-gather, three spectra in flight) against the spectrum computed whole on this rank's GPU,
and the two-phase shard extinction against the one-call form bit for bit at PB_STAGE_SPLIT=1;
the layer decomposition (all-to-all + all-gather, pipelined) and the stacked shards (two
atmospheres per submission, one all-gather of both) against the same spectrum."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['PB_STAGE_SPLIT'] = '1'


def main():
    import torch
    import torch.distributed as dist
    from pyratbay_amd import engine, synth
    from pyratbay_amd import dist as pbdist

    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', rank))
    case = synth.lbl_case(3001, 14, 12000, wnosamp=24, nlor=20, ndop=10, extent=80.0,
                          cutoff=4.0, niso=2, seed=11)
    whole = engine.LBLSpectrum(case, rt_path='transit')
    want = whole.run().clone()
    out = {}

    pipe = pbdist.ShardPipeline(case, world, rank, depth=3, kmax_exchange=True,
                                voigt=whole.voigt, lines=whole.lines, timestamps=False)
    fulls = []
    for i in range(7):
        r = pipe.submit()
        if r is not None:
            fulls.append(r[0].clone())
    fulls.append(pipe.flush()[0].clone())
    torch.cuda.synchronize()
    err = max(float(torch.max(torch.abs(f / want - 1)).item()) for f in fulls)
    out['wavenumber_vs_single'] = err
    # two-phase shard (maxima all-reduced) == one-call shard (maxima over every line), bit for bit
    g = pipe.gathers[0]
    one = engine.LBLSpectrum(case, rt_path='transit', voigt=whole.voigt, lines=whole.lines,
                             wbegin=g.wbegin, wcount=g.wcount)
    one.run()
    two = pipe.models[0]
    out['two_phase_equals_one_call'] = bool(torch.equal(one.ec, two.ec))

    sharded = pbdist.LayerShardedTransit(case, world, rank, voigt=whole.voigt, lines=whole.lines)
    outs = []
    for _ in range(5):
        r = sharded.submit()
        if r is not None:
            outs.append(r.clone())
    outs += [o.clone() for o in sharded.flush()]
    torch.cuda.synchronize()
    out['layers_vs_single'] = max(float(torch.max(torch.abs(o / want - 1)).item()) for o in outs)

    # the stacked form bench.py --gpus N runs: 2 atmospheres per submission, one all-reduce of their
    # maxima, ONE all-gather of their shards (dist.StackGather), two submissions in flight
    spipe = pbdist.ShardPipeline(case, world, rank, depth=2, kmax_exchange=True,
                                 voigt=whole.voigt, lines=whole.lines, stack=2)
    stacked = []
    for i in range(4):
        r = spipe.submit()
        if r is not None:
            stacked += [x.clone() for x in r[0]]
    stacked += [x.clone() for x in spipe.flush()[0]]
    torch.cuda.synchronize()
    assert len(stacked) == 8
    out['stacked_vs_single'] = max(float(torch.max(torch.abs(f / want - 1)).item())
                                   for f in stacked)

    worst = torch.tensor([out['wavenumber_vs_single'],
                          max(out['layers_vs_single'], out['stacked_vs_single']),
                          0.0 if out['two_phase_equals_one_call'] else 1.0],
                         dtype=torch.float64, device='cuda')
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        w = worst.cpu().tolist()
        print(json.dumps({'wavenumber_vs_single': w[0], 'layers_vs_single': w[1],
                          'two_phase_equals_one_call': w[2] == 0.0}), flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
