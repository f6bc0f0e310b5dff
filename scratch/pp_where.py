import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(0)
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, h, w_, cout = 1, 8, 32, 256
nch = cin // 32
x = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
wf, _ = K.prep_weights(w, True, False)
y = K.conv2d_fprop(x, wf, None, (h, w_), cout, 3, 0, 1.0, None).float()
wb = w.to(torch.bfloat16).float()
xf = x.float()
basis, names = [], []
for t in range(9):
    for a in range(nch):
        for b in range(nch):
            wt = torch.zeros_like(wb)
            # pixel chunk a multiplied with weight chunk b of tap t: put weight chunk b at input-channel position a
            wt[t // 3, t % 3, a * 32:(a + 1) * 32, :] = wb[t // 3, t % 3, b * 32:(b + 1) * 32, :]
            basis.append(F.conv2d(xf.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1).reshape(-1))
            names.append((t, a, b))
A = torch.stack(basis, 1).double().cpu()
sol = torch.linalg.lstsq(A, y.reshape(-1, 1).double().cpu()).solution.view(-1)
res = (A @ sol.view(-1, 1) - y.reshape(-1, 1).double().cpu()).abs().max().item()
print("residual of fit", res)
for nm, c in zip(names, sol.tolist()):
    if abs(c) > 0.02: print("tap %d pixel-chunk %d weight-chunk %d : %.3f" % (nm + (c,)))
