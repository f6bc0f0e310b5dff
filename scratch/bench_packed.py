"""filter gradient of the critic's 3-channel input convs (conv_wgrad_packed_kernel<true>): warm timing"""
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
for (h, ks) in ((32, 3), (16, 1)):
    x = torch.randn(128, h, h, 3, device=dev).to(K.BF16)
    dy = torch.randn(128, h, h, 128, device=dev).to(K.BF16)
    dw = torch.zeros(ks, ks, 3, 128, device=dev)
    db = torch.zeros(128, device=dev)
    t = warm(lambda: K.conv2d_wgrad(x, dy, dw, (h, h), ks, dbias=db))
    print(f'wgrad 3->128 k={ks} {h}x{h} n=128: {t:6.1f} us', flush=True)
