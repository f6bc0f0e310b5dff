"""What D.Block.2.Conv1 (256 -> 256 at 16x16, n = 128) would cost with the spatially constant label half of its input factored out:
forward 128 -> 256, input gradient 256 -> 128, filter gradient 128 x 256 -- warm-loop event timings of the existing kernels."""
import torch
from gan_lib_tensorflow_amd import kernels as K

def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

N = 128
g = torch.Generator().manual_seed(0)
for cin, cout in ((256, 256), (128, 256), (256, 128)):
    x = torch.randn((N, 16, 16, cin), generator=g).to(K.BF16).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) * 0.03).cuda()
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
    bias = torch.zeros(cout, device="cuda")
    t = timeit(lambda: K.img16_conv3x3(x, rf, bias, cout, K.IN_RELU))
    print(f"img16 {cin:3d} -> {cout:3d}: {t:6.1f} us")
for cin, cout in ((256, 256), (128, 256)):
    x = torch.randn((N, 16, 16, cin), generator=g).to(K.BF16).cuda()
    dy = torch.randn((N, 16, 16, cout), generator=g).to(K.BF16).cuda()
    dw = torch.zeros((3, 3, cin, cout), device="cuda")
    jobs = []
    def f():
        jobs.clear()
        K.conv2d_wgrad(x, dy, dw, (16, 16), 3, K.IN_RELU, slab_jobs=jobs)
        K.sum_slabs(jobs)
    t = timeit(f)
    print(f"wgrad {cin:3d} x {cout:3d}: {t:6.1f} us (incl. its slab sum)")
