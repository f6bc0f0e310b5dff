"""in-graph timing (one hipGraph of `reps` launches, HIP events around the replay) of the critic's batched 8x8 filter gradient -- the
eager loop of bench_critic_wgrad.py is bound by the host's launch rate for kernels this small.  TUNING library knobs: GANK_WGRAD_DBG."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
n, reps = 128, 20
def rnd(*shape):
    return torch.randn(*shape, device=dev).to(K.BF16)
items = [(rnd(n, 8, 8, 128), rnd(n, 8, 8, 128), torch.zeros(3, 3, 128, 128, device=dev), torch.zeros(128, device=dev)) for _ in range(4)]
def fn():
    K.conv2d_wgrad_batched(items, (8, 8), 3, K.IN_RELU)
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
print(f"batched 4 x (8x8, 128->128) n={n}: {sorted(ts)[len(ts)//2]:.1f} us per launch in a graph (min {min(ts):.1f})  env: " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith('GANK_')))
