"""CPU, gloo, world_size=2: the N>1 exchange path.  Each rank computes critic gradients on ITS shard with
the oracle graph, writes them into the flat gradient buffer of a ParamStore, the package's all-reduce sums
the buffers and Adam's grad_scale=1/world averages -- which must equal the gradient of the mean of the
per-shard losses computed in one process (SURVEY 8e parity statement)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gan_lib_tensorflow_amd import parallel
from gan_lib_tensorflow_amd.store import ParamStore
from oracle import ref_torch as T

B = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_inputs(rank):
    rng = np.random.default_rng(100 + rank)
    z = torch.tensor(rng.normal(size=(B, 128)))
    labels = torch.tensor(rng.integers(0, 10, B))
    real = torch.tensor(rng.normal(size=(B, 3072)) * 0.5)
    return z, labels, real


def _local_grads(P, rank):
    z, labels, real = _shard_inputs(rank)
    loss, _, _ = T.d_loss_fn(P, None, labels, z, None, towers=1, real_pre=real)
    names = T.trainable_names(P, 'Discriminator')
    return names, torch.autograd.grad(loss, [P[k] for k in names]), loss


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    pg, r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(2)
    state = T.init_sngan_params(0)
    P = T.to_torch(state)
    names, grads, _ = _local_grads(P, rank)
    store = ParamStore("cpu")
    for k in names:                      # same names / shapes as the product store
        store.get_variable(k, None, state[k])
    flat = store.flatten('Discriminator')
    with torch.no_grad():
        for k, g in zip(names, grads):
            store.vars[k].main_grad.copy_(g.to(torch.float32))
    parallel.allreduce_sum_(flat["grads"], pg)
    assert parallel.data_seed(0, 0) != parallel.data_seed(0, 1)
    if rank == 0:
        out.put({k: (store.vars[k].main_grad / world).numpy().copy() for k in names})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_gradient_of_mean_loss():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    P = T.to_torch(T.init_sngan_params(0))
    losses = [_local_grads_loss(P, r) for r in range(2)]
    loss = sum(losses) / 2
    names = T.trainable_names(P, 'Discriminator')
    ref = torch.autograd.grad(loss, [P[k] for k in names])
    for k, g in zip(names, ref):
        np.testing.assert_allclose(got[k], g.numpy(), rtol=2e-5, atol=1e-9, err_msg=k)


def _local_grads_loss(P, rank):
    z, labels, real = _shard_inputs(rank)
    return T.d_loss_fn(P, None, labels, z, None, towers=1, real_pre=real)[0]


def test_single_rank_is_a_noop():
    t = torch.ones(5)
    assert parallel.allreduce_sum_(t, None) is t
    assert parallel.init_from_env() == (None, 0, 1) or os.environ.get("WORLD_SIZE", "1") != "1"
