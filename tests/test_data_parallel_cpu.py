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


# ---- bucketed exchange of the generator's gradients ---------------------------------------------------------------
def _g_inputs(rank):
    rng = np.random.default_rng(300 + rank)
    return torch.tensor(rng.normal(size=(2 * B, 128))), torch.tensor(rng.integers(0, 10, 2 * B))


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    pg, r, w = parallel.init_from_env(backend="gloo")
    torch.set_num_threads(2)
    from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import G_BUCKETS
    state = T.init_sngan_params(0)
    P = T.to_torch(state)
    z, fl = _g_inputs(rank)
    loss, _ = T.g_loss_fn(P, z, fl, towers=1)
    names = T.trainable_names(P, 'Generator')
    grads = torch.autograd.grad(loss, [P[k] for k in names])
    store = ParamStore("cpu")
    for k in names:
        store.get_variable(k, None, state[k])
    flat = store.flatten('Generator')
    with torch.no_grad():
        for k, g in zip(names, grads):
            store.vars[k].main_grad.copy_(g.to(torch.float32))
    ranges = parallel.bucket_ranges(flat, G_BUCKETS)
    gb = parallel.GradBuckets(flat["grads"], ranges, pg)
    before = flat["grads"].clone()
    seen = []
    for k in reversed(range(len(ranges))):       # the order the backward pass finishes them: output side first
        gb.launch(k)
        # buckets not launched yet still hold the LOCAL gradients: nothing outside bucket k was touched
        for j, (a, b_) in enumerate(ranges):
            if j < k:
                assert torch.equal(flat["grads"][a:b_], before[a:b_]), (k, j)
        seen.append(k)
    gb.join()
    if rank == 0:
        out.put(({k: (store.vars[k].main_grad / world).numpy().copy() for k in names}, ranges, flat["params"].numel()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_generator_allreduce_equals_gradient_of_mean_loss_per_bucket():
    """parallel.bucket_ranges tiles the generator's flat gradient buffer block by block (SNGAN G_BUCKETS) and
    parallel.GradBuckets all-reduces the buckets one at a time, output side first; after the last one every bucket holds
    the gradient of the mean of the two ranks' losses (the reference averages its tower losses, gan_cifar_resnet.py:498)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got, ranges, total = out.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert len(ranges) == 4 and ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    sizes = [b - a for a, b in ranges]
    assert sizes[0] > sizes[1] and sizes[3] < 20000          # G.Input + G.Block.1 is the big one; G.OutputNorm + G.Output is tiny
    P = T.to_torch(T.init_sngan_params(0))
    loss = sum(T.g_loss_fn(P, *_g_inputs(r), towers=1)[0] for r in range(2)) / 2
    names = T.trainable_names(P, 'Generator')
    ref = torch.autograd.grad(loss, [P[k] for k in names])
    for k, g in zip(names, ref):
        np.testing.assert_allclose(got[k], g.numpy(), rtol=2e-5, atol=1e-9, err_msg=k)


def test_bucket_ranges_rejects_bad_groupings():
    store = ParamStore("cpu")
    for k in ("Net/A/w", "Net/B/w", "Net/A/b"):
        store.get_variable(k, None, np.zeros(5, np.float32))
    flat = store.flatten("Net")
    import pytest
    with pytest.raises(ValueError, match="not contiguous"):
        parallel.bucket_ranges(flat, (("A/",), ("B/",)))          # A's variables are not adjacent
    with pytest.raises(ValueError, match="belongs to 0"):
        parallel.bucket_ranges(flat, (("A/",),))
    assert parallel.bucket_ranges(flat, (("Net/",),)) == [(0, flat["params"].numel())]


def test_watchdog_drain_is_a_noop_without_an_rccl_group():
    """parallel.drain_collective_watchdog (called before every hipGraph capture): returns at once when there is no process group
    or the group is not RCCL -- the gloo rehearsals and single-process runs must not pay its sleep"""
    import time
    from gan_lib_tensorflow_amd import parallel
    t0 = time.perf_counter()
    parallel.drain_collective_watchdog(seconds=5.0)
    assert time.perf_counter() - t0 < 1.0
