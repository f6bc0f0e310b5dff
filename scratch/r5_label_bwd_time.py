"""warm-loop timing of the factored label conv's small kernels at the critic's shapes (n = 128, 16x16, 128 + 128 -> 256)"""
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

g = torch.Generator().manual_seed(0)
N = 128
dy = torch.randn((N, 16, 16, 256), generator=g).to(K.BF16).cuda()
w = (torch.randn((3, 3, 256, 256), generator=g) * 0.03).cuda()
T = torch.randn((10, 128), generator=g).to(K.BF16).cuda()
labels = torch.randint(0, 10, (N,), generator=g, dtype=torch.int32).cuda()
dw = torch.zeros_like(w)
bias = torch.zeros(256, device="cuda")
print(f"table {timeit(lambda: K.label_conv3x3_table(w, 128, T, bias, labels)):6.1f} us")
_, lists = K.label_conv3x3_table(w, 128, T, bias, labels)
print(f"bwd (tap sums + label kernel) {timeit(lambda: K.label_conv3x3_bwd(dy, lists, T, w, 128, dw)):6.1f} us")
