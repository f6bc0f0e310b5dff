"""Check the two-group LDS-DMA conv kernel (GANK_IGEMM_PP=1) against a torch fp32 convolution, then time it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lib_tensorflow_amd import kernels as K

def ref(x, w, b, relu, res):
    xf = x.float()
    if relu: xf = xf.relu()
    y = F.conv2d(xf.permute(0, 3, 1, 2), w.to(torch.bfloat16).float().permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
    if res is not None: y = y + res.float()
    return y

torch.manual_seed(0)
bad = 0
for (n, h, w_, cin, cout, relu, use_res) in [(1, 8, 32, 32, 256, 0, 0), (2, 8, 32, 64, 256, 1, 1), (3, 16, 32, 96, 256, 0, 1), (2, 32, 32, 256, 256, 0, 0),
                                             (2, 32, 64, 64, 512, 1, 0), (5, 32, 32, 128, 256, 0, 1)]:
    x = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
    b = torch.randn(cout, device="cuda")
    res = torch.randn(n, h, w_, cout, device="cuda").to(torch.bfloat16) if use_res else None
    wf, _ = K.prep_weights(w, True, False)
    for rep in range(3):
        y = K.conv2d_fprop(x, wf, b, (h, w_), cout, 3, K.IN_RELU if relu else 0, 1.0, res)
        torch.cuda.synchronize()
        yr = ref(x, w, b, relu, res)
        err = (y.float() - yr).abs().max().item()
        tol = 0.02 * yr.abs().max().item()
        nbad = ((y.float() - yr).abs() > tol).sum().item()
        print(f"n{n} {h}x{w_} {cin}->{cout} relu{relu} res{use_res} rep{rep}: max err {err:.4f} (tol {tol:.4f}) bad {nbad}", flush=True)
        bad += nbad
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
