"""dump the fused chain's outputs and input gradient for one configuration (GANK_RES8_CFG with the -DGANK_TUNING build), or compare two dumps:
    GANK_LIB_NAME=libgank_tune.so GANK_RES8_CFG=12 python scratch/res8_compare.py dump /tmp/a.pt ; ... =122 ... dump /tmp/b.pt ; python scratch/res8_compare.py cmp /tmp/a.pt /tmp/b.pt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        print(f"{k:12s} max |d| {float(d.max()):.3e} of max {float(a[k].double().abs().max()):.3e}; elements differing {int((d > 0).sum())} / {d.numel()}; "
              f"beyond 1.5e-2 of max: {int((d > 1.5e-2 * a[k].double().abs().max()).sum())}")
    sys.exit(0)
from gan_lib_tensorflow_amd import kernels as K
from gan_lib_tensorflow_amd import functional as Fn
out = {}
for (n, nb, pool) in ((2, 2, False), (3, 2, True), (128, 2, True)):
    rng = np.random.default_rng(800 + n + 10 * nb + pool)
    x = torch.tensor(rng.normal(size=(n, 8, 8, 128)).astype(np.float32)).to(torch.bfloat16).cuda().requires_grad_(True)
    params = []
    for b in range(nb):
        blk = []
        for j in range(2):
            w = torch.tensor((rng.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4).astype(np.float32)).to(torch.bfloat16).float().cuda().requires_grad_(True)
            blk += [w, torch.tensor((rng.normal(size=128) * 0.1).astype(np.float32)).cuda().requires_grad_(True)]
        params.append(tuple(blk))
    K.prep_weights_batched([p[i] for p in params for i in (0, 2)], want_d=True, kinds=[4] * (2 * nb))
    y = Fn.res_chain8(x, params, pool=pool)
    g = torch.tensor(rng.normal(size=tuple(y.shape)).astype(np.float32)).to(torch.bfloat16).cuda()
    y.backward(g)
    torch.cuda.synchronize()
    out[f"y{n}"] = y.detach().float().cpu(); out[f"dx{n}"] = x.grad.float().cpu(); out[f"dw{n}"] = params[0][0].grad.float().cpu()
torch.save(out, sys.argv[2])
