"""Worker of tests/test_model_gpu.py::test_bucketed_generator_update_through_rccl_matches_the_one_piece_update: the data-parallel
generator / critic updates through a world-size-1 RCCL group, in a process of its own -- an abort inside RCCL's teardown
(observed once in `destroy_process_group` on a pool box, after every check had passed) must cost one test, not the pytest session.
Teardown is ORDERED: every trainer, captured graph and tensor that refers to the communicator is released first, the device is
synchronised (no collective or graph replay in flight), then the group is destroyed -- and only then `RCCL PATH OK` is printed;
the parent test fails on any non-zero exit code, so an abort anywhere (checks, teardown, interpreter shutdown) fails the suite."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import socket
import torch.distributed as dist
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the segmented backward pass produces the gradient buffer of the one-piece pass (same kernels, same order per tensor)
# (batch-norm statistics from the three-pass kernels here: sums accumulated by conv epilogues arrive in a different
# order on every run, and the generator gradient -- ill-conditioned under bf16, see test_d_and_g_gradients_vs_oracle --
# then differs by 2-5 % between two runs of the SAME pass (scratch/g_repro.py), which would hide a segmentation error;
# kept off for the trajectory comparison below as well, whose bounds were set on atomics-order noise of the backward pass alone)
from gan_lib_tensorflow_amd import functional as Fn
stats_were, Fn.CONV_EPILOGUE_STATS = Fn.CONV_EPILOGUE_STATS, False
tr = S.SNGANTrainer(batch_size=8, seed=17, use_graphs=False)
rng0 = tr.rng_state.clone()
tr._g_forward_backward()
whole = tr.g_flat["grads"].clone()
tr.rng_state.copy_(rng0)
phases, after = tr._g_phases()
assert after == [None, 3, 2, 1, 0, None]
for ph in phases[:-1]:
    ph()
torch.cuda.synchronize()
seg = tr.g_flat["grads"]
# (not bit-identical: the fp32 atomics of the batch-norm backward sums land in a different order on every run, and a
# gradient that crosses a bf16 rounding boundary is amplified on the way down -- two runs of the SAME pass differ alike)
assert float(whole.norm()) > 0 and float((seg - whole).norm() / whole.norm()) < 5e-3, float((seg - whole).norm() / whole.norm())
del tr
from gan_lib_tensorflow_amd import parallel as _parallel
_parallel.disable_collective_event_cache()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ok = False
try:
    results = {}
    # the exchange call under capture.  NOTE what this does and does not show: at world size 1 RCCL's all-reduce is a no-op,
    # so the captured graph holds NO collective node (torch warns "The CUDA Graph is empty") -- this only checks that
    # issuing the call inside a capture neither raises nor disturbs the buffer.  Replay of a real multi-rank collective
    # is unverified (no multi-GPU box), which is why SNGANTrainer captures collectives by default only at world size 1.
    buf = torch.randn(4099, device="cuda")
    want = buf.clone()
    dist.all_reduce(want)
    cap = buf.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(cap.clone())                       # warm-up outside the capture (communicator setup)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    from gan_lib_tensorflow_amd import parallel
    parallel.drain_collective_watchdog()               # the eager warm-up's work items leave the watchdog's list first
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dist.all_reduce(cap)
    cap.copy_(buf)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap, want)
    for name, pg, graphs, kw in (("plain", None, True, {}), ("bucketed", dist.group.WORLD, True, {}),
                                 ("bucketed_split", dist.group.WORLD, True, {"capture_collectives": False}),
                                 ("bucketed_wire16", dist.group.WORLD, True, {"grad_wire_dtype": "bf16"}),
                                 ("bucketed_eager", dist.group.WORLD, False, {})):
        tr = S.SNGANTrainer(batch_size=8, seed=17, use_graphs=graphs, process_group=pg, **kw)
        assert tr.bucketed == (pg is not None)
        feed = S.synthetic_batches(8, "cuda", seed=5)
        for _ in range(4):             # iterations 1.. run the generator update: eager, capture, two replays
            tr.train_iteration(feed)
        torch.cuda.synchronize()
        assert tr.use_graphs == graphs
        if name in ("bucketed", "bucketed_wire16"):
            # the collective CALLS sit inside the captured updates (world size 1: they add no node): ONE graph for the critic
            # update (no optimiser graph behind an eager all-reduce), ONE for the generator update with its four bucket
            # all-reduce calls on the communication stream
            assert tr.capture_collectives and isinstance(tr._graphs['g_seg'], torch.cuda.CUDAGraph) and 'g' not in tr._graphs
            assert tr._graphs['d_pre'][1] is None
        if name == "bucketed_split":
            assert len(tr._graphs['g_seg']) == 6 and 'g' not in tr._graphs        # forward, 4 segments, optimiser
            assert tr._graphs['d_pre'][1] is not None                             # graph / eager all-reduce / graph
        if name.startswith("bucketed"):
            assert [b.numel() for b in tr._g_buckets.buckets] == [b - a for a, b in
                                                                  __import__('gan_lib_tensorflow_amd').parallel.bucket_ranges(tr.g_flat, S.G_BUCKETS)]
        results[name] = (tr.g_flat["params"].clone(), tr.d_flat["params"].clone(), float(tr.g_loss), tr.rng_state.clone(), int(tr.g_opt.t))
        del tr, feed
        gc.collect()
    ref = results["plain"]
    for name in ("bucketed", "bucketed_split", "bucketed_wire16", "bucketed_eager"):
        got = results[name]
        assert torch.equal(got[3], ref[3]) and got[4] == ref[4] == 3
        # four iterations of TF-Adam (beta1 = 0) on trajectories that differ only by atomics ordering: see
        # test_train_steps_eager_vs_graph_and_oracle_update for why a few weights may differ by ~lr
        for a, b_ in ((got[0], ref[0]), (got[1], ref[1])):
            d = (a - b_).abs()
            assert torch.isfinite(a).all() and d.max().item() < 40 * 2e-4 and d.mean().item() < 3e-4, (name, d.max().item(), d.mean().item())      # measured 1.7e-4 .. 2.1e-4 over repeated runs
        assert abs(got[2] - ref[2]) < 0.5
    ok = True
finally:
    Fn.CONV_EPILOGUE_STATS = stats_were
    # ordered teardown: graphs that hold captured collective calls and tensors registered with the communicator go first,
    # then nothing may be in flight when the communicator is destroyed
    results = ref = got = g = cap = buf = want = side = None
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
if ok:
    print("RCCL PATH OK", flush=True)


