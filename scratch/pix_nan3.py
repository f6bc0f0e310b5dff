"""forward / backward NaN hunt in the Pix2Pix U-Net at batch 16: every conv2d_general / instance-norm result checked as it is produced"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get("POISON", "0") == "1":
    _empty, _empty_like = torch.empty, torch.empty_like
    def empty(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t
    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t
    torch.empty, torch.empty_like = empty, empty_like
from gan_lib_tensorflow_amd import functional as Fn, kernels as K
from gan_lib_tensorflow_amd.Pix2Pix.train import Pix2PixTrainer, default_args
orig = {}
def wrap(mod, name):
    f = getattr(mod, name)
    orig[name] = f
    def w(*a, **kw):
        out = f(*a, **kw)
        t = out[0] if isinstance(out, tuple) else out
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(t.float()).all())
        shapes = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]
        print(f"   {name:24s} {shapes} -> {tuple(t.shape)} finite {fin}" + ("" if fin else f"  NON-FINITE {int((~torch.isfinite(t.float())).sum())}"), flush=True)
        return out
    setattr(mod, name, w)
for nm in ("conv2d_fprop", "conv2d_general_fprop", "tap_gather_up2", "depth_to_space2", "cbn_fwd", "dropout_fwd", "concat_channels",
           "conv2d_dgrad", "conv2d_general_dgrad", "space_to_depth2", "cbn_bwd", "tap_scatter_up2", "split_channels", "dropout_bwd", "relu_bwd"):
    if hasattr(K, nm):
        wrap(K, nm)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tr = Pix2PixTrainer(default_args(batch_size=batch, crop_size=512, max_steps=1000), seed=13)
g = torch.Generator().manual_seed(9)
a = (torch.rand(batch, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
b = (torch.rand(batch, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
print("forward")
out = tr._generator(a)
print("backward")
tr._backward(Fn.l1_loss(out, b))
torch.cuda.synchronize()
bad = [k for k in tr.g_flat['names'] if not bool(torch.isfinite(tr.store.vars[k].main_grad).all())]
print("non-finite gradients:", bad[-6:], len(bad))
