import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lib_tensorflow_amd import kernels as K
n, h, cin, cout = 128, 32, 256, 256
x = torch.randn(n, h, h, cin, device="cuda").to(torch.bfloat16)
w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
wf, _ = K.prep_weights(w, True, False)
for _ in range(3):
    y = K.conv2d_fprop(x, wf, None, (h, h), cout, 3)
torch.cuda.synchronize()
st = y.view(-1).view(torch.int32)[:2048].cpu().view(2, 256, 4).long() & 0xFFFFFFFF
t0 = st[0, 0, 0].item()
for g in range(2):
    print("group", g, "(stamps: 0 start of read phase, 1 before B1, 2 after B1+lgkm, 3 after MFMA issue)")
    for p in list(range(0, 12)) + list(range(60, 72)):
        a = [(st[g, p, k].item() - t0) & 0xFFFFFFFF for k in range(4)]
        nxt = (st[g, p + 1, 0].item() - t0) & 0xFFFFFFFF if p + 1 < 72 else 0
        print(f"  p{p:2d}: read {a[1]-a[0]:5d}  barrier+lgkm {a[2]-a[1]:5d}  mfma {a[3]-a[2]:5d}  B2 {nxt-a[3]:6d}   phase total {nxt-a[0]:6d}   abs {a[0]}")
