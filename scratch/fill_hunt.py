"""Which Python call sites launch fill kernels during one eager training iteration? (autograd-materialised zero gradients
do not pass through Python: the difference to the profiler's FillFunctor count is theirs)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
tr = S.SNGANTrainer(batch_size=64)
feed = S.synthetic_batches(64, device="cuda") if hasattr(S, "synthetic_batches") else None
tr.use_graphs = False
tr.train_iteration(feed); tr.train_iteration(feed)
torch.cuda.synchronize()
hits = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        hits[(name, " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st)))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for nm in ("zeros", "zeros_like", "full", "full_like", "ones", "ones_like"):
    wrap(torch, nm)
for nm in ("zero_", "fill_"):
    wrap(torch.Tensor, nm)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_iteration(feed)
    torch.cuda.synchronize()
for k, v in hits.most_common():
    print(v, k)
n_fill = sum(e.count for e in prof.key_averages() if "FillFunctor" in e.key)
print("FillFunctor kernels in this iteration:", n_fill)
for e in prof.key_averages():
    if "fill" in e.key.lower() or "zero" in e.key.lower():
        print("  ", e.key[:80], e.count)
