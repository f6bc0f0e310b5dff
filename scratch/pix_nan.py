"""which generator gradients of the Pix2Pix U-Net are non-finite at batch 16 (512 x 512), under dispatcher switches"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get("POISON", "0") == "1":        # every torch.empty buffer starts as NaN: a kernel that reads what nobody wrote shows up at any batch size
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
def run(batch, tag):
    tr = Pix2PixTrainer(default_args(batch_size=batch, crop_size=512, max_steps=1000), seed=13)
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(batch, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    b = (torch.rand(batch, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    out = tr._generator(a)
    tr._backward(Fn.l1_loss(out, b))
    torch.cuda.synchronize()
    bad = {k: int((~torch.isfinite(tr.store.vars[k].main_grad)).sum()) for k in tr.g_flat['names'] if not bool(torch.isfinite(tr.store.vars[k].main_grad).all())}
    print(tag, "batch", batch, "output finite", bool(torch.isfinite(out.float()).all()), "non-finite gradients:", bad, flush=True)
    del tr
for bsz in (int(v) for v in (sys.argv[1:] or ["16"])):
    run(bsz, "default")
    Fn.PHASE_STACK_UPCONV4 = False
    run(bsz, "no phase stack")
    Fn.PHASE_STACK_UPCONV4 = True
