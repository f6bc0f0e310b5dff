"""warm timing of the LDS-patch kernel's launches (128-wide cout tiles: the critic's 16x16 layers, the generator's 16x16 input gradients)"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
def warm(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
for (n, h, cin, cout) in ((128, 16, 256, 128), (128, 16, 128, 128), (128, 32, 128, 128), (128, 16, 256, 256)):
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.02
    wf = K.prep_weights(w)[0]
    x = torch.randn(n, h, h, cin, device=dev).to(K.BF16)
    bias = torch.zeros(cout, device=dev)
    os.environ.get('X')
    t = warm(lambda: K.conv2d_fprop(x, wf, bias, (h, h), cout, 3, K.IN_RELU))
    print(f'conv {cin}->{cout} {h}x{h} n={n}: {t:7.1f} us ({n*h*h*cin*cout*18/t/1e6:6.0f} TF)', flush=True)
