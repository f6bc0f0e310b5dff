"""Data-parallel exchange for the train step: one process per GPU, `torch.distributed` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference's only parallelism is in-graph tower replication with averaged tower losses
(SNGAN/gan_cifar_resnet.py:324-332,436,498,523-526).  Here every rank is one tower-pair replica: it
accumulates its gradients into the network's flat fp32 gradient buffer, the buffers are summed with ONE
all-reduce per update (D: 1.70 M floats, G: 7.88 M floats -- few, large collectives suit the
point-to-point xGMI mesh) and the 1/world factor is applied inside the Adam kernel (`grad_scale`).
No exchange is needed for the spectral-norm `u` vectors (deterministic on identical weights) nor for
conditional-batch-norm statistics (per tower in the reference, normalization.py:47)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).  Returns
    (process_group | None, rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return None, 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    if not dist.is_initialized():
        dist.init_process_group(backend, **kw)
    return dist.group.WORLD, rank, world


def allreduce_sum_(flat_grads, group=None):
    """In-place SUM of a flat gradient buffer over the ranks of `group` (no-op for a single rank)."""
    if group is None or dist.get_world_size(group) == 1:
        return flat_grads
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def data_seed(base_seed, rank):
    """Parameter init uses `base_seed` on every rank (identical replicas without a broadcast);
    the data-side RNG (z, fake labels, dequantisation) must differ per rank."""
    return 1234567 + 7919 * int(rank) + int(base_seed)
