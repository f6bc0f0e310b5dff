"""warm timing of the two-group kernel's launches (plain 32x32, plain 16x16, phase 16->32), with and without statistics"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
wph, _ = K.upconv3x3_prep(w)
wf = K.prep_weights(w)[0]
bias = torch.zeros(256, device=dev)

def warm(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000

for n in (128, 320):
    x16 = torch.randn(n, 16, 16, 256, device=dev).to(K.BF16)
    x32 = torch.randn(n, 32, 32, 256, device=dev).to(K.BF16)
    for sg in (0, 2):
        t1 = warm(lambda: K.upconv3x3_fprop(x16, wph, bias, 256, stats_groups=sg))
        t2 = warm(lambda: K.conv2d_fprop(x32, wf, bias, (32, 32), 256, 3, stats_groups=sg))
        t3 = warm(lambda: K.conv2d_fprop(x16, wf, bias, (16, 16), 256, 3, stats_groups=sg))
        print(f'n={n} stats={sg}: upconv16->32 {t1:7.1f} us ({n*1024*256*1024*2/t1/1e6:6.0f} TF) | conv32 {t2:7.1f} us ({n*1024*256*2304*2/t2/1e6:6.0f} TF) | conv16 {t3:7.1f} us ({n*256*256*2304*2/t3/1e6:6.0f} TF)', flush=True)
