import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lib_tensorflow_amd import kernels as K
which = sys.argv[1] if len(sys.argv) > 1 else "fprop"
n, h, cin, cout, k = (int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (64, 32, 256, 256, 3)))
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
x = torch.randn(n, h, h, cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(k, k, cin, cout, device="cuda") / (k * k * cin) ** 0.5)
xh = torch.randn(n, h // 2, h // 2, cin, device="cuda").to(torch.bfloat16)
dy = torch.randn(n, h, h, cout, device="cuda").to(torch.bfloat16)
dyp = torch.randn(n, h // 2, h // 2, cout, device="cuda").to(torch.bfloat16)
wf, wd = K.prep_weights(w, True, True)
if os.environ.get("MICRO_FRAG", "0") == "1":
    (wf, wd), = K.prep_weights_batched([w], want_d=True, kinds=[3])
rfrag = K.prep_weights_batched([w], want_d=True, kinds=[4])[0] if (k == 3 and cin % 32 == 0 and cout % 32 == 0) else None
wup = K.upconv3x3_prep(w) if k == 3 and cin % 64 == 0 else None
dw = torch.zeros_like(w)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run():
    if which == "fprop":
        K.conv2d_fprop(x, wf, None, (h, h), cout, k)
    elif which == "upfprop":      # phase-decomposed UpsampleConv 3x3: x low-res [n,h/2,h/2,cin] -> [n,h,h,cout]
        K.upconv3x3_fprop(xh, wup[0], None, cout)
    elif which == "fprop_up":
        K.conv2d_fprop(xh, wf, None, (h, h), cout, k, K.IN_UPSAMPLE2X)
    elif which == "img16":        # 16x16 image-resident conv (gank_img16_conv3x3), h must be 16
        K.img16_conv3x3(x, rfrag[0], None, cout, K.IN_RELU)
    elif which == "dgrad":
        K.conv2d_dgrad(dy, wd, (h, h), cin, k)
    elif which == "cpwgrad":      # ConvMeanPool 3x3 filter gradient: x [n,h,h,cin], dy pooled [n,h/2,h/2,cout]
        K.convpool3x3_wgrad(x, dyp, dw, K.IN_RELU)
    elif which == "wgrad_relu":
        K.conv2d_wgrad(x, dy, dw, (h, h), k, K.IN_RELU)
    else:
        K.conv2d_wgrad(x, dy, dw, (h, h), k)
for _ in range(3): run()
torch.cuda.synchronize()
ev0.record()
for _ in range(reps): run()
ev1.record(); torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 1e3 / reps
fl = 2.0 * n * h * h * cin * cout * k * k
print(f"{which} n{n} h{h} {cin}->{cout} k{k}: {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s")
