"""Step times of the other configurations (BASELINE.json configs 3-5) on one GPU, eager execution, synthetic inputs.
usage: other_configs_bench.py [acgan|pggan|pix2pix ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches

from gan_lib_tensorflow_amd import kernels as K

which = sys.argv[1:] or ["acgan", "pggan", "pix2pix"]
PEAK = 2.5e15      # dense bf16 MFMA, MI355X


def conv_flops_per_step(make, step):
    """conv FLOPs one step executes (counted by the launchers themselves: gank_prof_*), from an EAGER trainer of the same
    configuration; -> FLOPs.  The step time comes from the captured trainer."""
    tr = make(False)
    step(tr)
    torch.cuda.synchronize()
    K.prof_reset(); K.prof_enable(True)
    step(tr)
    torch.cuda.synchronize()
    K.prof_enable(False)
    fl = sum(K.prof_collect(f)[2] for f in (0, 1))
    K.prof_reset()
    del tr
    return fl


def frac(fl, t):
    return f"{fl / 1e9:8.1f} GFLOP/step as run = {fl / t / 1e12:6.1f} TFLOP/s = {fl / t / PEAK:.3f} of the bf16 MFMA peak"


def timed(fn, n=5, warm=3):       # warm-up covers the eager first execution and the capture of every graph
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if "acgan" in which or "acgan32" in which:
    from gan_lib_tensorflow_amd.ACGAN.train import ACGANTrainer
    for bs in ((32,) if "acgan32" in which else (32, 256)):
        feed = synthetic_batches(bs, "cuda", seed=2)
        it = [0]

        def step(trn):
            it[0] += 1
            trn.train_iteration(feed, it[0])
        fl = conv_flops_per_step(lambda g: ACGANTrainer(batch_size=bs, seed=1, use_graphs=g), step)
        tr = ACGANTrainer(batch_size=bs, seed=1)
        t = timed(lambda: step(tr))
        print(f"ACGAN   bs={bs:4d}: {1e3 * t:8.2f} ms per step (1 G + 5 D updates) = {5 * bs / t:9.0f} real images/s; {frac(fl, t)}", flush=True)
        del tr
if "pggan" in which:
    from gan_lib_tensorflow_amd.PGGAN.train import PGGANTrainer, default_args
    for bc, trans in ((3, False), (4, True), (6, False)):
        size = 4 * 2 ** bc
        feed = synthetic_batches(16, "cuda", seed=2)
        fl = conv_flops_per_step(lambda g: PGGANTrainer(default_args(batch_size=16, block_count=bc, image_size=size, trans=trans), seed=1, use_graphs=g),
                                 lambda trn: trn.train_iteration(feed))
        tr = PGGANTrainer(default_args(batch_size=16, block_count=bc, image_size=size, trans=trans), seed=1)
        t = timed(lambda: tr.train_iteration(feed), n=3)
        print(f"PGGAN   {size:3d}x{size:<3d} trans={int(trans)} bs=16: {1e3 * t:8.2f} ms per step (1 G + 5 D updates) = {5 * 16 / t:9.0f} real images/s; {frac(fl, t)}", flush=True)
        del tr
if "pix2pix" in which:
    from gan_lib_tensorflow_amd.Pix2Pix.train import Pix2PixTrainer, default_args
    for bs in (4, 16):
        g = torch.Generator().manual_seed(3)
        a = (torch.rand(bs, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
        b = (torch.rand(bs, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
        fl = conv_flops_per_step(lambda gr: Pix2PixTrainer(default_args(batch_size=bs, crop_size=512), seed=1, use_graphs=gr), lambda trn: trn.train_step(a, b))
        tr = Pix2PixTrainer(default_args(batch_size=bs, crop_size=512), seed=1)
        t = timed(lambda: tr.train_step(a, b), n=3)
        print(f"Pix2Pix 512x512 bs={bs:2d}: {1e3 * t:8.2f} ms per step (5 D + 1 G updates) = {bs / t:7.1f} pairs/s; {frac(fl, t)}", flush=True)
        del tr
