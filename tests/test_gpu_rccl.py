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
