"""a few isolated launches of the resident ConvMeanPool kernels (critic shapes) for rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd import kernels as K
for (n, hp, cin) in ((128, 16, 128), (128, 8, 256)):
    x = torch.randn(n, 2 * hp, 2 * hp, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, 3, cin, 128, device="cuda") / 34.
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[5])
    dy = torch.randn(n, hp, hp, 128, device="cuda").to(torch.bfloat16)
    b = torch.zeros(128, device="cuda")
    for _ in range(3):
        K.cpool_res_fprop(x, rf, b, 128, K.IN_RELU, dy)
        K.cpool_res_dgrad(dy, rd, cin, x)
    torch.cuda.synchronize()
