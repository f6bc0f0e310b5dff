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


def init_from_env(backend=None, device=None, force=False):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).  Returns
    (process_group | None, rank, world).  force: a single process still gets a (world-size-1) group, so the data-parallel
    code path -- bucketed gradients, collectives inside the captured updates -- can be run and timed on one GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and not force:
        return None, 0, 1
    if world == 1:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    disable_collective_event_cache()
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    if not dist.is_initialized():
        dist.init_process_group(backend, **kw)
    return dist.group.WORLD, rank, world


def allreduce_sum_(flat_grads, group=None, wire_dtype=None, single_rank_too=False):
    """In-place SUM of a flat gradient buffer over the ranks of `group` (no-op for a single rank unless single_rank_too:
    the world-size-1 rehearsal of the data-parallel path issues the collective all the same).
    wire_dtype = the 16-bit activation dtype: the buffer travels (and is summed by RCCL) in 16 bits -- half the bytes on the
    xGMI links for two cast launches; the fp32 buffer is rewritten from the summed copy."""
    if group is None or (dist.get_world_size(group) == 1 and not single_rank_too):
        return flat_grads
    if wire_dtype is not None and flat_grads.is_cuda:
        from . import kernels as K
        wire = K.to_bf16(flat_grads)
        dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=group)
        K.to_f32(wire, out=flat_grads)
        return flat_grads
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def data_seed(base_seed, rank):
    """Parameter init uses `base_seed` on every rank (identical replicas without a broadcast);
    the data-side RNG (z, fake labels, dequantisation) must differ per rank."""
    return 1234567 + 7919 * int(rank) + int(base_seed)


def bucket_ranges(flat, groups):
    """Contiguous [start, end) element ranges of a ParamStore.flatten() buffer, one per group of variable-name
    substrings (forward / creation order).  Every trainable variable of the buffer must fall in exactly one group and the
    members of a group must be adjacent in the buffer -- the layout flatten() produces for a network built block by block."""
    names, offsets = flat["names"], flat["offsets"]
    total = flat["params"].numel()
    ends = [offsets[names[i + 1]] if i + 1 < len(names) else total for i in range(len(names))]
    owner = []
    for k in names:
        hit = [gi for gi, subs in enumerate(groups) if any(sub in k for sub in subs)]
        if len(hit) != 1:
            raise ValueError(f"{k} belongs to {len(hit)} gradient buckets")
        owner.append(hit[0])
    ranges = []
    for gi in range(len(groups)):
        idx = [i for i, o in enumerate(owner) if o == gi]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError(f"bucket {gi} ({groups[gi]}) is empty or not contiguous in the flat buffer")
        ranges.append((offsets[names[idx[0]]], ends[idx[-1]]))
    if sorted(ranges) != ranges or ranges[0][0] != 0 or ranges[-1][1] != total or any(a[1] != b[0] for a, b in zip(ranges, ranges[1:])):
        raise ValueError(f"buckets do not tile the buffer: {ranges} of {total}")
    return ranges


def disable_collective_event_cache():
    """Call before `init_process_group("nccl")` in a process that will capture hipGraphs.  ProcessGroupNCCL recycles the events
    of its work items from a per-device cache; an event that once marked a collective issued UNDER CAPTURE can come back as the
    end event of an eager collective, and the watchdog thread's `hipEventQuery` on it is refused with `hipErrorCapturedEvent`
    (process-terminating, from the watchdog thread) while its old stream is capturing again.  With the cache off every work
    item creates its own events, so an eager collective's events have never been near a capture: the hazard itself is gone,
    not waited out.  (`drain_collective_watchdog` stays as a second line: it costs a synchronise + 0.25 s per capture, a
    handful per run.)  Remaining failure mode: a torch build that ignores TORCH_NCCL_CUDA_EVENT_CACHE -- then only the drain
    stands between an eager collective's pending query and the next capture, and that is a matter of timing."""
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")


def drain_collective_watchdog(seconds=0.25):
    """Call before a hipGraph capture in a process that owns an RCCL process group.  ProcessGroupNCCL's watchdog thread polls
    the end event of every EAGER collective (`hipEventQuery`, every 100 ms) until it has completed; those events are recycled
    from a cache that also served collectives issued under capture, and HIP refuses the query
    (`hipErrorCapturedEvent`, which terminates the process from the watchdog thread) while the stream such an event was once
    captured on is capturing again.  Observed once in 30-odd runs of tests/rccl_worker.py, right after an eager first execution
    was followed by a capture.  After a device synchronise every eager collective has completed, and two watchdog periods
    later none is left on its list; captures happen a handful of times per run."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and torch.cuda.is_available():
        import time
        torch.cuda.synchronize()
        time.sleep(seconds)


class GradBuckets:
    """The flat gradient buffer of a network cut into contiguous buckets that are all-reduced one by one, each as soon as
    the backward pass has produced its last gradient, on a communication stream of their own: the collective of bucket k
    (RCCL over xGMI; SUM, the 1/world factor is applied by the optimiser) runs beside the backward kernels of bucket k+1.
    The reference sums its tower gradients in one tf.add_n at the end of the graph (SNGAN/gan_cifar_resnet.py:523-526);
    bucketing changes the schedule, not the arithmetic: every element is summed over the same ranks exactly once."""

    def __init__(self, flat_grads, ranges, group, wire_dtype=None):
        self.group = group
        self.buckets = [flat_grads[a:b] for a, b in ranges]
        self.cuda = flat_grads.is_cuda
        self.comm = torch.cuda.Stream(device=flat_grads.device) if self.cuda else None
        self.wire_dtype = wire_dtype if self.cuda else None      # 16-bit buckets on the wire (allreduce_sum_)

    def launch(self, k):
        """all-reduce bucket k; call right after the last kernel that writes it was enqueued on the current stream.
        Under hipGraph capture the fork onto the communication stream and the join become edges of the captured graph: the
        collective is a node beside the next segment's kernels."""
        if self.group is None:
            return
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                allreduce_sum_(self.buckets[k], self.group, self.wire_dtype, single_rank_too=True)
        else:
            dist.all_reduce(self.buckets[k], op=dist.ReduceOp.SUM, group=self.group)

    def join(self):
        """the compute stream waits for every launched collective (before the optimiser reads the gradients)"""
        if self.cuda and self.group is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
