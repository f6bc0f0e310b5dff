"""warm timing of the generator's 1x1 shortcut convs (256 -> 256; the 4x4 one 1024 -> 256)"""
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
for (n, h, cin) in ((320, 16, 256), (320, 8, 256), (320, 4, 1024), (128, 16, 256), (128, 8, 256)):
    w = torch.randn(1, 1, cin, 256, device=dev) * 0.02
    wf = K.prep_weights(w)[0]
    x = torch.randn(n, h, h, cin, device=dev).to(K.BF16)
    bias = torch.zeros(256, device=dev)
    t = warm(lambda: K.conv2d_fprop(x, wf, bias, (h, h), 256, 1))
    mb = (x.numel() + n * h * h * 256) * 2 / 1e6
    print(f'1x1 {cin}->256 {h}x{h} n={n}: {t:7.1f} us  ({mb / t * 1e-6 * 1e6:6.2f} TB/s of in+out)', flush=True)
