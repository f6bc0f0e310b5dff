"""Worker of tests/test_data_parallel_gpu.py::test_ranks_agree_when_one_rank_cannot_capture_its_collectives: two data-parallel
ranks of the real trainer on cuda:0 (gloo rehearsal).  The trainer is told that its collectives can be captured
(`capture_collectives = True`, what an RCCL group gets with `--capture-collectives`), and the exchange call is made to FAIL
under capture on rank 1 only -- the situation `SNGANTrainer._agree_on_capture` exists for: rank 0 captures its update as one
graph, rank 1 cannot, and unless both take the split form (graph / eager all-reduce / graph) the ranks issue different
collective sequences and hang.  Both ranks must fall back together, keep training, and stay bit-identical replicas."""
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
    real_allreduce = parallel.allreduce_sum_
    seen = {"capture_calls": 0}

    def allreduce_under_test(flat_grads, group=None, wire_dtype=None, single_rank_too=False):
        if torch.cuda.is_current_stream_capturing():
            seen["capture_calls"] += 1
            if rank == 1:
                raise RuntimeError("forced: this rank cannot capture its collective")
            return flat_grads            # rank 0 "captures" the call (gloo itself must never run under capture)
        return real_allreduce(flat_grads, group, wire_dtype, single_rank_too)

    parallel.allreduce_sum_ = allreduce_under_test
    tr = S.SNGANTrainer(batch_size=16, device=device, seed=0, use_graphs=True, process_group=pg)
    assert tr.capture_collectives is False           # gloo: never by default
    tr.capture_collectives = True
    feed = S.synthetic_batches(16, device, seed=rank)
    tr.train_iteration(feed)                         # iteration 0: the critic update's capture fails on rank 1
    torch.cuda.synchronize()
    assert seen["capture_calls"] >= 1
    assert tr.capture_collectives is False, f"rank {rank} still believes in captured collectives"
    assert tr.use_graphs and tr._graphs['d_pre'][1] is not None, f"rank {rank}: not the split form"
    tr.capture_collectives = True                    # the same for the bucketed generator update (captured at iteration 1)
    before = seen["capture_calls"]
    tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert seen["capture_calls"] > before
    assert tr.capture_collectives is False and isinstance(tr._graphs['g_seg'], list) and len(tr._graphs['g_seg']) == 6
    for _ in range(2):                               # replays of the split forms, eager collectives between the graphs
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.use_graphs
    for flat in (tr.g_flat, tr.d_flat):
        p = flat["params"]
        assert bool(torch.isfinite(p).all())
        ref = p.clone()
        dist.broadcast(ref, src=0, group=pg)
        assert torch.equal(p, ref), float((p - ref).abs().max())
    del tr, feed
    torch.cuda.synchronize()
    dist.barrier(group=pg)
    dist.destroy_process_group()
    print(f"rank {rank} agreed", flush=True)


if __name__ == "__main__":
    main()
