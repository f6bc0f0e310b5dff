"""Which Python call sites launch fill / add kernels during one eager ACGAN training iteration?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd.ACGAN.train import ACGANTrainer
from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches
tr = ACGANTrainer(batch_size=32, seed=1, use_graphs=False)
feed = synthetic_batches(32, "cuda", seed=2)
tr.train_iteration(feed, 1); tr.train_iteration(feed, 2)
torch.cuda.synchronize()
hits = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=5)[:-1]
        hits[(name, " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st)))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for nm in ("zeros", "zeros_like", "full", "full_like", "ones", "ones_like"):
    wrap(torch, nm)
for nm in ("zero_", "fill_"):
    wrap(torch.Tensor, nm)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_iteration(feed, 3)
    torch.cuda.synchronize()
for k, v in hits.most_common(25):
    print(v, k)
for e in prof.key_averages():
    if "at::native" in e.key or "aten::add" in e.key or "aten::fill" in e.key or "aten::zero" in e.key:
        print("  ", e.key[:110], e.count)
