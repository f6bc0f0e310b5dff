"""stage-by-stage check of the fused 8x8 residual chain against torch-CPU float64"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gan_lib_tensorflow_amd import kernels as K
from oracle import ref_torch as T

def bf(a):
    t = torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)
    return t.to(torch.float64), t.cuda().contiguous()
def rel(a, b):
    a = a.detach().to(torch.float64).cpu(); b = b.detach().to(torch.float64)
    return round(float((a - b).norm() / b.norm()), 5)
rng = np.random.default_rng(0)
n = 3
x, xt = bf(rng.normal(size=(n, 8, 8, 128)))
w1, _ = bf(rng.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4)
w2, _ = bf(rng.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4)
w1t, w2t = w1.float().cuda(), w2.float().cuda()
K.prep_weights_batched([w1t, w2t], want_d=True, kinds=[4, 4])
out, h1s, ys = K.res8_chain_fwd(xt, [w1t._prep_res[0], w2t._prep_res[0]], [None, None], True, False)
h1 = T.conv2d_same(torch.relu(x), w1)
y = x + T.conv2d_same(torch.relu(h1), w2)
print("fwd h1", rel(h1s[0], h1), "y", rel(ys[0], y))
dy, dyt = bf(rng.normal(size=(n, 8, 8, 128)))
dx, g1s, dys = K.res8_chain_bwd(dyt, None, None, [w1t._prep_res[1], w2t._prep_res[1]], h1s, [xt], keep=True)
# oracle with the HIP path's own masks
h1m = (h1s[0].double().cpu() > 0).double()
xm = (x > 0).double()
w2f = torch.flip(w2, (0, 1)).permute(0, 1, 3, 2)
w1f = torch.flip(w1, (0, 1)).permute(0, 1, 3, 2)
dh2 = T.conv2d_same(dy, w2f)
g1 = dh2 * h1m
dh0 = T.conv2d_same(g1, w1f)
dxr = dy + dh0 * xm
print("bwd g1", rel(g1s[0], g1), "dx", rel(dx, dxr), "dh2-unmasked check", rel(g1s[0], dh2))

# ---- two blocks, pooled
ws = []
for i in range(4):
    w, _ = bf(rng.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4)
    ws.append(w)
wts = [w.float().cuda() for w in ws]
K.prep_weights_batched(wts, want_d=True, kinds=[4] * 4)
for pool in (False, True):
    out, h1s, ys = K.res8_chain_fwd(xt, [w._prep_res[0] for w in wts], [None] * 4, True, pool)
    xr = x.clone().requires_grad_(True)
    h1a = T.conv2d_same(torch.relu(xr), ws[0]); ya = xr + T.conv2d_same(torch.relu(h1a), ws[1])
    h1b = T.conv2d_same(torch.relu(ya), ws[2]); yb = ya + T.conv2d_same(torch.relu(h1b), ws[3])
    ref = torch.relu(yb).mean(dim=(1, 2)) if pool else yb
    print("pool", pool, "fwd", rel(h1s[0], h1a), rel(ys[0], ya), rel(h1s[1], h1b), rel(ys[1], yb), rel(out, ref))
    g, gt = bf(rng.normal(size=tuple(ref.shape)))
    for t in (h1a, ya, h1b, yb):
        t.retain_grad()
    ref.backward(g)
    dx, g1s, dys = K.res8_chain_bwd(None if pool else gt, gt if pool else None, ys[-1] if pool else None, [w._prep_res[1] for w in wts], h1s, [xt, ys[0]], keep=True)
    print("   bwd dy_b", rel(dys[1], yb.grad), "g1_b", rel(g1s[1], h1b.grad), "dy_a", rel(dys[0], ya.grad), "g1_a", rel(g1s[0], h1a.grad), "dx", rel(dx, xr.grad))
