"""The generator's 8x8 layers (256 -> 256, 3x3): the LDS-resident kernel against the kernels the dispatcher used before.
usage: python scratch/bench_res8g.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gan_lib_tensorflow_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
C = 256


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for n, groups in ((128, 2), (320, 10)):
    w = (torch.randn(3, 3, C, C, device=dev) / 48).float()
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
    wf, wd = K.prep_weights(w, True, True)
    wph, wd4 = K.upconv3x3_prep(w)
    b = torch.randn(C, device=dev)
    x8 = torch.randn(n, 8, 8, C, device=dev).to(K.BF16)
    x4 = torch.randn(n, 4, 4, C, device=dev).to(K.BF16)
    r4 = torch.randn(n, 4, 4, C, device=dev).to(K.BF16)
    dy = torch.randn(n, 8, 8, C, device=dev).to(K.BF16)
    if True:          # (no statistics arena: both sides pay their own zero fill)
        rows = [
            ("conv2 fprop + stats + half-res residual", lambda: K.conv2d_fprop(x8, wf, b, (8, 8), C, 3, K.RES_UPSAMPLE2X, 1.0, r4, stats_groups=groups),
             lambda: K.res8_conv3x3(x8, rf, b, C, K.RES_UPSAMPLE2X, r4, stats_groups=groups)),
            ("upconv 4->8 fprop + stats", lambda: K.upconv3x3_fprop(x4, wph, b, C, 0, None, stats_groups=groups),
             lambda: K.res8_conv3x3(x4, rf, b, C, K.IN_UPSAMPLE2X, None, stats_groups=groups)),
            ("conv2 dgrad", lambda: K.conv2d_dgrad(dy, wd, (8, 8), C, 3, 0, 1.0), lambda: K.res8_conv3x3(dy, rd, None, C, 0)),
            ("upconv 4->8 dgrad", lambda: K.upconv3x3_dgrad(dy, wd4, C), lambda: K.res8_conv3x3(dy, rd, None, C, K.OUT_POOLSUM2X)),
        ]
        for name, old, new in rows:
            print("n=%3d  %-42s %7.1f us -> %7.1f us" % (n, name, timeit(old), timeit(new)), flush=True)
