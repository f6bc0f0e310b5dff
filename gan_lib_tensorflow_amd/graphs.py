"""hipGraph capture-and-replay of a train-step piece (torch.cuda.CUDAGraph = hipGraph on ROCm).

A GAN update at the reference's batch sizes is a few hundred kernel launches of 5-50 us; issued eagerly from Python they
cost ~100 us of host time each and the GPU idles between them (ACGAN at 32 samples per GPU: 69 ms per step eager).  Each
piece runs eagerly once (allocator warm-up, lazy initialisation; this IS the step), is captured on the next use and
replayed from then on.  What a captured piece may depend on: device memory at fixed addresses only -- inputs are copied into
static buffers, step counters / learning rates / fade-in weights / RNG state live on the device and are updated OUTSIDE the
captured region.  A capture that fails RAISES (the eager first execution already was that step, so nothing is lost): a run
that asked for graphs must not silently become a 10x slower eager run.  `allow_eager_fallback=True` restores the degradation
to eager execution with a message on stderr.
"""
import gc
import sys

import torch


class GraphRunner:
    def __init__(self, enabled=True, allow_eager_fallback=False):
        self.enabled = enabled and torch.cuda.is_available()
        self.allow_eager_fallback = allow_eager_fallback
        self.graphs = {}
        self._seen = set()

    def run(self, key, fn):
        """fn(): enqueues kernels only (no host synchronisation, no host-dependent control flow that changes between calls)"""
        if not self.enabled:
            return fn()
        g = self.graphs.get(key)
        if g is not None:
            g.replay()
            return None
        if key not in self._seen:
            # first use: eager, on a side stream (so that lazily created state is not tied to the capture)
            self._seen.add(key)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                out = fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            return out
        try:
            g = torch.cuda.CUDAGraph()
            was = gc.isenabled()
            from . import parallel
            parallel.drain_collective_watchdog()   # no eager collective left on the RCCL watchdog's list when a capture starts
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                gc.disable()             # a cyclic-GC pass that frees another graph's pool memory mid-capture aborts the process
                try:
                    fn()
                finally:
                    if was:
                        gc.enable()
            self.graphs[key] = g
            g.replay()                   # capture executed nothing: this is the step
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize()
            if not self.allow_eager_fallback:
                raise RuntimeError(f"hipGraph capture of {key!r} failed ({e}); pass allow_eager_fallback=True (or use_graphs=False) "
                                   f"to run eagerly") from e
            print(f"[gank] hipGraph capture of {key!r} failed ({e}); running eagerly", file=sys.stderr)
            self.enabled = False
            return fn()
        return None

    def clear(self):
        self.graphs.clear()
        self._seen.clear()
