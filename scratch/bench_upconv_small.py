"""micro-benchmark: the generator's low-resolution UpsampleConv launches, warm (back to back) and in step-like conditions
(after an MFMA-heavy kernel, caches flushed)"""
import sys, torch
sys.path.insert(0, '.')
from gan_lib_tensorflow_amd import kernels as K

dev = torch.device('cuda')
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
wph, _ = K.upconv3x3_prep(w)
wf = K.conv3x3_prep(w)[0] if hasattr(K, 'conv3x3_prep') else None
bias = torch.zeros(256, device=dev)
big_x = torch.randn(128, 16, 16, 256, device=dev).to(K.BF16)
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)

def warm(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000

def cold(fn, heavy, do_flush, reps=20):
    ts = []
    for _ in range(reps):
        if heavy:
            for _ in range(6): K.upconv3x3_fprop(big_x, wph, bias, 256)
        if do_flush: flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000)
    ts.sort()
    return ts[len(ts) // 2]

for n, hl in ((320, 4), (128, 4), (320, 8)):
    x = torch.randn(n, hl, hl, 256, device=dev).to(K.BF16)
    sg = 10 if n == 320 else 2
    fn = lambda: K.upconv3x3_fprop(x, wph, bias, 256, flags=K.IN_RELU, stats_groups=sg)
    print(f'upconv n={n} {hl}x{hl}: warm {warm(fn):6.1f} us | event-pair alone {cold(lambda: None, False, False):5.1f} | '
          f'flushed {cold(fn, False, True):6.1f} | after heavy {cold(fn, True, False):6.1f} | heavy+flushed {cold(fn, True, True):6.1f}', flush=True)
