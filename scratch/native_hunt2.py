"""Python stacks behind aten::fill_ / zero_ / copy_ / clone of one eager training iteration (torch.profiler with_stack).
usage: native_hunt2.py <pix2pix|pggan|acgan>"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches
which = sys.argv[1]
if which == "pix2pix":
    from gan_lib_tensorflow_amd.Pix2Pix.train import Pix2PixTrainer, default_args
    tr = Pix2PixTrainer(default_args(batch_size=4, crop_size=512), seed=1, use_graphs=False)
    a = torch.randn(4, 512, 512, 3, device="cuda").to(torch.bfloat16); b = torch.randn(4, 512, 512, 3, device="cuda").to(torch.bfloat16)
    step = lambda: tr.train_step(a, b)
elif which == "pggan":
    from gan_lib_tensorflow_amd.PGGAN.train import PGGANTrainer, default_args
    tr = PGGANTrainer(default_args(batch_size=16, block_count=4, image_size=64, trans=True), seed=1, use_graphs=False)
    feed = synthetic_batches(16, "cuda", seed=2)
    step = lambda: tr.train_iteration(feed)
else:
    from gan_lib_tensorflow_amd.ACGAN.train import ACGANTrainer
    tr = ACGANTrainer(batch_size=32, seed=1, use_graphs=False)
    feed = synthetic_batches(32, "cuda", seed=2)
    it = [0]
    def step():
        it[0] += 1; tr.train_iteration(feed, it[0])
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::mul") and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "gan_lib_tensorflow_amd" in s or "torch/autograd" in s or "scratch" in s][:4]
        agg[(e.name, str(e.input_shapes)[:80], " <- ".join(s.split("/")[-1][:60] for s in st))] += 1
for k, v in agg.most_common(40):
    print(v, k)
