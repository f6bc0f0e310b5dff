"""Check the two-group LDS-DMA conv kernel (GANK_IGEMM_PP=2: all forms) against torch fp32 convolutions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gan_lib_tensorflow_amd import kernels as K

def conv_ref(x, w, b, relu, res):
    xf = x.float()
    if relu: xf = xf.relu()
    y = F.conv2d(xf.permute(0, 3, 1, 2), w.to(torch.bfloat16).float().permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
    if res is not None: y = y + res.float()
    return y

def report(name, y, yr):
    err = (y.float() - yr).abs()
    tol = 0.02 * yr.abs().max().item()
    nbad = (err > tol).sum().item()
    print(f"{name}: max err {err.max().item():.4f} (tol {tol:.4f}) bad {nbad}", flush=True)
    return nbad

torch.manual_seed(0)
bad = 0
for (n, h, w_, cin, cout, relu, use_res) in [(2, 8, 32, 64, 256, 1, 1), (2, 32, 32, 256, 256, 0, 0), (2, 32, 64, 64, 512, 1, 0), (5, 32, 32, 128, 256, 0, 1),
                                             (3, 16, 16, 64, 256, 0, 1), (2, 16, 16, 256, 256, 1, 0), (1, 32, 16, 128, 256, 0, 0), (2, 16, 48, 64, 256, 0, 0)]:
    x = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
    b = torch.randn(cout, device="cuda")
    res = torch.randn(n, h, w_, cout, device="cuda").to(torch.bfloat16) if use_res else None
    wf, wd = K.prep_weights(w, True, True)
    for rep in range(2):
        y = K.conv2d_fprop(x, wf, b, (h, w_), cout, 3, K.IN_RELU if relu else 0, 1.0, res)
        bad += report(f"fprop n{n} {h}x{w_} {cin}->{cout} relu{relu} res{use_res} rep{rep}", y, conv_ref(x, w, b, relu, res))
    if cin % 256 == 0:   # dgrad = conv of dy with the flipped, transposed filter; relu mask in the epilogue
        dy = torch.randn(n, h, w_, cout, device="cuda").to(torch.bfloat16)
        ref_in = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
        dx = K.conv2d_dgrad(dy, wd, (h, w_), cin, 3, 0, 1.0, None, ref_in)
        wt = w.to(torch.bfloat16).float().flip(0, 1).permute(2, 3, 0, 1)          # [Cin, Cout, kh, kw]
        dxr = F.conv2d(dy.float().permute(0, 3, 1, 2), wt, None, padding=1).permute(0, 2, 3, 1) * (ref_in.float() > 0)
        bad += report(f"dgrad n{n} {h}x{w_} {cout}->{cin}", dx, dxr)
# phase form: UpsampleConv 3x3 fprop (low-res grid h x w_) and ConvMeanPool 3x3 input gradient
for (n, h, w_, cin, cout) in [(2, 16, 16, 64, 256), (3, 16, 16, 256, 256), (2, 8, 32, 128, 256), (1, 16, 32, 64, 512)]:
    x = torch.randn(n, h, w_, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
    b = torch.randn(cout, device="cuda")
    wph, wd4 = K.upconv3x3_prep(w)
    y = K.upconv3x3_fprop(x, wph, b, cout)
    xu = x.float().repeat_interleave(2, 1).repeat_interleave(2, 2)
    yr = F.conv2d(xu.permute(0, 3, 1, 2), w.float().permute(3, 2, 0, 1), b, padding=1).permute(0, 2, 3, 1)
    err = (y.float() - yr).abs(); tol = 0.03 * yr.abs().max().item(); nb = (err > tol).sum().item()
    print(f"upconv n{n} {h}x{w_} {cin}->{cout}: max err {err.max().item():.4f} (tol {tol:.4f}) bad {nb}", flush=True)
    bad += nb
    # ConvMeanPool dgrad: dy pooled [n,h,w_,cout2] -> dx [n,2h,2w_,cin2] with cin2 % 256 == 0
    cin2, cout2 = cout, cin
    w2 = torch.randn(3, 3, cin2, cout2, device="cuda") / (9 * cin2) ** 0.5
    wp4, wphd = K.convpool3x3_prep(w2)
    dyp = torch.randn(n, h, w_, cout2, device="cuda").to(torch.bfloat16)
    dx = K.convpool3x3_dgrad(dyp, wphd, cin2)
    xx = torch.zeros(n, 2 * h, 2 * w_, cin2, device="cuda", requires_grad=True)
    yy = F.avg_pool2d(F.conv2d(xx.permute(0, 3, 1, 2), w2.permute(3, 2, 0, 1), None, padding=1), 2).permute(0, 2, 3, 1)
    yy.backward(dyp.float())
    err = (dx.float() - xx.grad).abs(); tol = 0.03 * xx.grad.abs().max().item(); nb = (err > tol).sum().item()
    print(f"convpool dgrad n{n} {h}x{w_} {cout2}->{cin2}: max err {err.max().item():.4f} (tol {tol:.4f}) bad {nb}", flush=True)
    bad += nb
print("TOTAL BAD", bad)
sys.exit(1 if bad else 0)
