"""warm timing of G.Output (256 -> 3, 3x3, 32x32, tanh): forward (plain and with the fused CBN + relu staging), input gradient, filter gradient"""
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
    x = torch.randn(n, 32, 32, 256, device=dev).to(K.BF16)
    w = torch.randn(3, 3, 256, 3, device=dev) / 48.
    b = torch.zeros(3, device=dev)
    wf, wd = K.prep_weights(w, True, True)
    dy = torch.randn(n, 32, 32, 3, device=dev).to(K.BF16)
    dw = torch.zeros_like(w)
    t1 = warm(lambda: K.conv2d_fprop(x, wf, b, (32, 32), 3, 3, K.OUT_TANH))
    groups = n // 32
    labels = torch.randint(0, 10, (n,), device=dev, dtype=torch.int32)
    gamma, beta = torch.ones(10, 256, device=dev), torch.zeros(10, 256, device=dev)
    stats = torch.cat([torch.zeros(groups, 1, 256, device=dev), torch.ones(groups, 1, 256, device=dev)], 1).contiguous()
    t2 = warm(lambda: K.cbn_relu_conv3x3_fprop(x, labels, gamma, beta, stats, wf, b, 3, K.OUT_TANH))
    t3 = warm(lambda: K.conv2d_dgrad(dy, wd, (32, 32), 256, 3))
    t4 = warm(lambda: K.conv2d_wgrad(x, dy, dw, (32, 32), 3, 0, 1.0, dbias=b))
    mb = n * 1024 * 256 * 2 / 1e6
    print(f"n={n}: fprop {t1:6.1f} us ({mb/t1:5.2f} TB/s of x), with CBN + relu staging {t2:6.1f} us | dgrad {t3:6.1f} us ({mb/t3:5.2f} TB/s of dx) | wgrad {t4:6.1f} us ({mb/t4:5.2f} TB/s of x)", flush=True)
