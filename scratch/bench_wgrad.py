"""warm timing of the generator's 256 -> 256 3x3 filter gradients (all-taps kernel)"""
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
for h in (32, 16):
    x = torch.randn(128, h, h, 256, device=dev).to(K.BF16)
    dy = torch.randn(128, h, h, 256, device=dev).to(K.BF16)
    dw = torch.zeros(3, 3, 256, 256, device=dev)
    for fl in (0, K.IN_RELU):
        t = warm(lambda: K.conv2d_wgrad(x, dy, dw, (h, h), 3, fl))
        print(f'wgrad 256->256 3x3 {h}x{h} n=128 flags={fl}: {t:7.1f} us ({128*h*h*256*2304*2/t/1e6:6.0f} TF)', flush=True)
