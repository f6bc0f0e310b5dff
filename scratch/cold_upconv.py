"""G.Block.1.Conv1 (4x4 -> 8x8 UpsampleConv 3x3, 1024 -> 256) alone under rocprofv3 --kernel-trace: hot, after a rewrite of its
operand / its input / both, after a 1-GB fill, with and without the statistics epilogue; as a graph replay (no host gaps)."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
mode, n, stats = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(n, 4, 4, 1024, device=dev).to(K.BF16); x2 = x.clone()
w = torch.randn(3, 3, 1024, 256, device=dev) / 96.
wup = K.upconv3x3_prep(w)
w0 = wup[0]; w2 = w0.clone()
b = torch.zeros(256, device=dev)
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)
def body():
    for _ in range(10):
        if mode in ("w", "wx"): w0.copy_(w2)
        if mode in ("x", "wx"): x.copy_(x2)
        if mode == "evict": big.fill_(1.0)
        K.upconv3x3_fprop(x, w0, b, 256, K.IN_RELU, None, 2 if stats else 0)
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(3): g.replay()
torch.cuda.synchronize()
