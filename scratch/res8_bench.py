"""time the fused 8x8 residual chain (fwd, bwd) at the critic's shape; GANK_RES8_CFG picks the kernel variant"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd import kernels as K
n = int(os.environ.get('RES8_N', '128'))
x = torch.randn(n, 8, 8, 128, device="cuda").to(torch.bfloat16)
ws = [(torch.randn(3, 3, 128, 128, device="cuda") / 34.) for _ in range(4)]
K.prep_weights_batched(ws, want_d=True, kinds=[4] * 4)
bs = [torch.zeros(128, device="cuda") for _ in range(4)]
dp = torch.randn(n, 128, device="cuda").to(torch.bfloat16)
def fwd(): return K.res8_chain_fwd(x, [w._prep_res[0] for w in ws], bs, True, True)
out, h1s, ys = fwd()
def bwd(keep): return K.res8_chain_bwd(None, dp, ys[-1], [w._prep_res[1] for w in ws], h1s, [x, ys[0]], keep=keep)
def timeit(f, reps=300):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
print("N", n, "cfg", os.environ.get("GANK_RES8_CFG", "default"), "fwd %.1f us  bwd(train) %.1f us  bwd(dx only) %.1f us" % (timeit(fwd), timeit(lambda: bwd(True)), timeit(lambda: bwd(False))))
