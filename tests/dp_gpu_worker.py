"""Worker of tests/test_data_parallel_gpu.py: one data-parallel rank of the real trainer (hipGraph fwd+bwd ->
all-reduce -> hipGraph Adam).  Launched by torch.distributed.run; every rank uses cuda:0 (one-GPU rehearsal, gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_lib_tensorflow_amd import parallel  # noqa: E402
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    pg, rank, world = parallel.init_from_env(backend="gloo")
    assert world == 2
    tr = S.SNGANTrainer(batch_size=16, device=device, seed=0, use_graphs=True, process_group=pg)
    feed = S.synthetic_batches(16, device, seed=rank)
    for _ in range(3):
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.use_graphs, "graph capture fell back to eager"
    for flat in (tr.g_flat, tr.d_flat):
        p = flat["params"]
        assert bool(torch.isfinite(p).all())
        ref = p.clone()
        dist.broadcast(ref, src=0, group=pg)
        # identical init + summed gradients + identical optimiser => identical replicas (bit for bit: the all-reduce
        # result is the same tensor on every rank)
        assert torch.equal(p, ref), float((p - ref).abs().max())
    # the ranks drew different data: their last critic losses differ
    losses = [torch.zeros(1, device=device) for _ in range(world)]
    dist.all_gather(losses, tr.d_loss.clone(), group=pg)
    assert float((losses[0] - losses[1]).abs()) > 0
    dist.barrier(group=pg)
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
