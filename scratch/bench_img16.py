"""warm timing of the 16x16 image-resident conv (gank_img16_conv3x3) against the kernels the dispatcher used before"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
def warm(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
for n in (128, 320):
    x = torch.randn(n, 16, 16, 256, device=dev).to(K.BF16)
    w = torch.randn(3, 3, 256, 256, device=dev) / 48.
    b = torch.zeros(256, device=dev)
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
    wf, wd = K.prep_weights(w, True, True)
    fl = 2 * n * 256 * 256 * 2304
    t1 = warm(lambda: K.img16_conv3x3(x, rf, b, 256, K.IN_RELU))
    t2 = warm(lambda: K.conv2d_fprop(x, wf, b, (16, 16), 256, 3, K.IN_RELU))
    t3 = warm(lambda: K.img16_conv3x3(x, rd, None, 256, 0, relu_ref=x))
    t4 = warm(lambda: K.conv2d_dgrad(x, wd, (16, 16), 256, 3, 0, 1.0, None, x))
    print(f"n={n} 256->256 16x16: fprop resident {t1:6.1f} us ({fl/t1/1e6:5.0f} TF) | igemm {t2:6.1f} us ({fl/t2/1e6:5.0f} TF) || dgrad resident {t3:6.1f} us | igemm {t4:6.1f} us", flush=True)
# the critic's layer: 128 -> 128 at n = 128 (128 whole-image workgroups; the half-image form fills the chip)
for n in (64, 128, 256):
    x = torch.randn(n, 16, 16, 128, device=dev).to(K.BF16)
    w = torch.randn(3, 3, 128, 128, device=dev) / 34.
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
    t1 = warm(lambda: K.img16_conv3x3(x, rf, None, 128, K.IN_RELU))
    t3 = warm(lambda: K.img16_conv3x3(x, rd, None, 128, 0, relu_ref=x))
    print(f"n={n} 128->128 16x16: fprop resident {t1:6.1f} us | dgrad resident {t3:6.1f} us", flush=True)
