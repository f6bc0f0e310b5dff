"""in-graph timing of gank_cpool_res_dgrad_image_wgrad (D.Block.1: 128 x 16x16 pooled, 128 channels) against the two launches it replaces.
TUNING library knob: GANK_IMGWG_DBG (1 = no final atomics, 2 = no filter-gradient section, 4 = no Xcol build)."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
n, reps = 128, 20
def rnd(*shape):
    return torch.randn(*shape, device=dev).to(K.BF16)
img, h1, dy, pooled = rnd(n, 32, 32, 3), rnd(n, 32, 32, 128), rnd(n, 16, 16, 128), rnd(n, 16, 16, 3)
w2 = torch.randn(3, 3, 128, 128, device=dev) / 34.
(rf, rd), = K.prep_weights_batched([w2], want_d=True, kinds=[5])
big = torch.zeros(257 * 8192, device=dev)
dw1, db1, dws, dbs = big[:27 * 128].view(3, 3, 3, 128), torch.zeros(128, device=dev), torch.zeros(1, 1, 3, 128, device=dev), torch.zeros(128, device=dev)
def fused():
    K.cpool_res_dgrad_image_wgrad(dy, rd, h1, img, dw1, db1, pooled, dws, dbs)
def two():
    dx = K.cpool_res_dgrad(dy, rd, 128, h1)
    K.conv2d_wgrad_narrow_pair((img, dx, dw1, db1, (32, 32), 3), (pooled, dy, dws, dbs, (16, 16), 1))
def dgrad_only():
    K.cpool_res_dgrad(dy, rd, 128, h1)
def timeit(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / reps * 1000)
    return sorted(ts)[len(ts) // 2]
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith('GANK_') and k != 'GANK_LIB_NAME')
print(f"fused {timeit(fused):6.1f} us   two launches {timeit(two):6.1f} us   input gradient alone {timeit(dgrad_only):6.1f} us   [{env}]", flush=True)
