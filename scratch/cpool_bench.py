"""time the resident ConvMeanPool kernels at the critic's two shapes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd import kernels as K
def timeit(f, reps=200):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for (n, hp, cin) in ((128, 16, 128), (128, 8, 256)):
    x = torch.randn(n, 2 * hp, 2 * hp, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, 3, cin, 128, device="cuda") / 34.
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[5])
    wp4, wphd = K.convpool3x3_prep(w)
    dy = torch.randn(n, hp, hp, 128, device="cuda").to(torch.bfloat16)
    b = torch.zeros(128, device="cuda")
    print(f"N{n} pooled {hp}x{hp} {cin}->128: resident fprop %.1f dgrad %.1f | igemm fprop %.1f dgrad %.1f us" % (
        timeit(lambda: K.cpool_res_fprop(x, rf, b, 128, K.IN_RELU, dy)), timeit(lambda: K.cpool_res_dgrad(dy, rd, cin, x)),
        timeit(lambda: K.convpool3x3_fprop(x, wp4, b, 128, K.IN_RELU, dy)), timeit(lambda: K.convpool3x3_dgrad(dy, wphd, cin, x))))
