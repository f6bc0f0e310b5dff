"""Which Python call sites are behind the framework kernels (fill / copy / add / mul) of one eager training iteration?
usage: native_hunt.py <pix2pix|pggan|acgan>"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
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
try:
    step(); step()
except TypeError as e:
    print("signature:", e); raise
torch.cuda.synchronize()
hits = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        hits[(name, " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st)))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for nm in ("zeros", "zeros_like", "full", "full_like", "ones", "ones_like", "cat", "tensor"):
    wrap(torch, nm)
for nm in ("zero_", "fill_", "copy_", "clone", "contiguous", "to", "__mul__", "__rmul__", "__add__", "__radd__", "add_", "mul_", "float", "__sub__", "__neg__", "__truediv__"):
    wrap(torch.Tensor, nm)
step()
torch.cuda.synchronize()
for k, v in hits.most_common(45):
    print(v, k)
