"""One-GPU rehearsal of the N > 1 path of the real trainer: 2 ranks (gloo all-reduce of the flat gradient buffers,
both on cuda:0) must keep bit-identical replicas through captured updates.  (RCCL itself needs one GPU per rank: the
driver's multi-GPU bench exercises it; the exchange arithmetic is covered on CPU by test_data_parallel_cpu.py.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_trainer_replicas_stay_identical():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


@pytest.mark.gpu
def test_ranks_agree_when_one_rank_cannot_capture_its_collectives():
    """`SNGANTrainer._agree_on_capture` (MIN over ranks): rank 1's capture of the exchange call is forced to fail, rank 0's
    succeeds -- both must end in the split form (graph / eager all-reduce / graph) for the critic update AND the bucketed
    generator update, neither may hang (timeout), and the replicas stay bit-identical afterwards (tests/dp_agree_worker.py)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_agree_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 agreed" in r.stdout and "rank 1 agreed" in r.stdout
