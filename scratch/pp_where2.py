import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(0)
cin = 64
n, h, w_, cout = 1, 8, 32, 256
x = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
wf, _ = K.prep_weights(w, True, False)
y = K.conv2d_fprop(x, wf, None, (h, w_), cout, 3, 0, 1.0, None).float()
yr = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float().permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1)
bad = (y - yr).abs() > 0.05
print("bad by ch%32", bad.view(n, h, w_, cout // 32, 32).sum(dim=(0, 1, 2, 3)).tolist())
# does y at channel c equal yr at some other channel?
yy = y[0, 3, 5]; rr = yr[0, 3, 5]
for c in range(0, 32):
    d = (rr - yy[c]).abs()
    j = int(d.argmin())
    print(c, "matches ref channel", j, "err %.3f" % d[j].item())
