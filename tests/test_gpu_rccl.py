"""The collectives of pyratbay_amd/dist.py through RCCL itself (backend 'nccl'), in a ONE-rank
process group on cuda:0 -- what a one-GPU box can show of the multi-GPU path: torch's
ProcessGroupNCCL accepts every tensor the sharded run hands it (the int64 alias of the
library's maxima buffer included), issued from side streams, and the stream ordering holds
(results equal the collective-free run to 1e-12).  The data movement between ranks is covered
by the gloo tests (tests/test_dist_gloo.py) and by bench.py's parity check before timing.
Runs in a child process so that the process group does not outlive the test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_collectives_through_rccl_one_rank():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0',
               WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    run = subprocess.run([sys.executable, os.path.join(HERE, 'rccl_one_rank.py')], env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    res = json.loads(run.stdout.strip().splitlines()[-1])
    assert res['backend'] == 'nccl'
    assert res['two_phase_vs_one_call'] <= 1e-12
    assert res['layer_pipeline_vs_single'] <= 1e-12
    assert res['shard_pipeline_vs_single'] <= 1e-12


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()          # (does not initialise the GPU)
    except Exception:                              # noqa: BLE001
        return 0


@pytest.mark.skipif(_gpu_count() < 2, reason='needs two GPUs (the round-end box has one)')
def test_two_ranks_over_rccl():
    """Two ranks on two GPUs over RCCL (runs wherever a node offers them): the wavenumber
    decomposition as bench.py runs it against the single-GPU spectrum (1e-10), the two-phase shard
    extinction against the one-call form bit for bit (PB_STAGE_SPLIT=1), the layer decomposition
    against the same spectrum."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r),
                   WORLD_SIZE='2', LOCAL_RANK=str(r),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'rccl_two_ranks.py')],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res['wavenumber_vs_single'] <= 1e-10
    assert res['layers_vs_single'] <= 1e-10
    assert res['two_phase_equals_one_call']
