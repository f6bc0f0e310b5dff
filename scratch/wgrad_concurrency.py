"""Upper bound for fusing the critic's filter gradients into fewer launches: the same eight launches of one critic update
(batch 128) back to back on ONE stream against round-robin over several streams (no graphs, no dependencies).
usage: python scratch/wgrad_concurrency.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_lib_tensorflow_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
N = 128
g = torch.Generator(device="cpu").manual_seed(0)


def t(*shape):
    return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)


def z(*shape):
    return torch.zeros(shape, dtype=torch.float32, device=dev)


jobs = []
# D.Block.1.Conv2: ConvMeanPool 3x3 128->128, x 32x32
x, dy, dw, db = t(N, 32, 32, 128), t(N, 16, 16, 128), z(3, 3, 128, 128), z(128)
jobs.append(("cpool 32->16", lambda x=x, dy=dy, dw=dw, db=db: K.convpool3x3_wgrad(x, dy, dw, K.IN_RELU, dbias=db)))
# D.Block.2.Conv2: ConvMeanPool 3x3 128->128, x 16x16
x, dy, dw, db = t(N, 16, 16, 128), t(N, 8, 8, 128), z(3, 3, 128, 128), z(128)
jobs.append(("cpool 16->8", lambda x=x, dy=dy, dw=dw, db=db: K.convpool3x3_wgrad(x, dy, dw, K.IN_RELU, dbias=db)))
# D.Block.2.Conv1: 3x3 256->128 at 16x16
x, dy, dw, db = t(N, 16, 16, 256), t(N, 16, 16, 128), z(3, 3, 256, 128), z(128)
jobs.append(("taps 16x16 256->128", lambda x=x, dy=dy, dw=dw, db=db: K.conv2d_wgrad(x, dy, dw, (16, 16), 3, K.IN_RELU, 1.0, dbias=db)))
# D.Block.2 shortcut: 1x1 256->128 at 8x8
x, dy, dw, db = t(N, 8, 8, 256), t(N, 8, 8, 128), z(1, 1, 256, 128), z(128)
jobs.append(("1x1 8x8 256->128", lambda x=x, dy=dy, dw=dw, db=db: K.conv2d_wgrad(x, dy, dw, (8, 8), 1, 0, 1.0, dbias=db)))
# D.Block.3/4: four 3x3 128->128 at 8x8, one launch
items = [(t(N, 8, 8, 128), t(N, 8, 8, 128), z(3, 3, 128, 128), z(128)) for _ in range(4)]
jobs.append(("rows x4 8x8", lambda items=items: K.conv2d_wgrad_batched(items, (8, 8), 3, K.IN_RELU, 1.0)))
# D.Block.1.Conv1 (3x3 3->128 at 32x32) + shortcut (1x1 3->128 at 16x16): the streaming pair
a = (t(N, 32, 32, 3), t(N, 32, 32, 128), z(3, 3, 3, 128), z(128), (32, 32), 3)
b = (t(N, 16, 16, 3), t(N, 16, 16, 128), z(1, 1, 3, 128), z(128), (16, 16), 1)
jobs.append(("narrow pair", lambda a=a, b=b: K.conv2d_wgrad_narrow_pair(a, b)))


def run(streams, reps):
    main = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(main)
    for _ in range(reps):
        for st in streams:
            st.wait_stream(main)
        for i, (_, fn) in enumerate(jobs):
            with torch.cuda.stream(streams[i % len(streams)]):
                fn()
        for st in streams:
            main.wait_stream(st)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, fn in jobs:          # each alone
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-24s %7.1f us alone (back to back with itself)" % (name, e0.elapsed_time(e1) * 1e3 / 50), flush=True)

main = torch.cuda.current_stream()
for ns in (1, 2, 3, 6):
    streams = [main] if ns == 1 else [torch.cuda.Stream() for _ in range(ns)]
    run(streams, 5)
    print("%d stream(s): %7.1f us per set of %d launches" % (ns, min(run(streams, 40) for _ in range(3)), len(jobs)), flush=True)
