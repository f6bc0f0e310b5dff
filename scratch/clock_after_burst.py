"""Does a small kernel run slower right after a burst of heavy MFMA work (clock still throttled)?  Under rocprofv3 --kernel-trace:
10 x [ 20 launches of the 256 -> 256 3x3 conv at 32x32, n = 320 (6 ms) ; 6 launches of the 16x16 image kernel ] and, for comparison,
the image kernel alone with the same host pacing."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
burst = int(sys.argv[1])
n = 128
x = torch.randn(n, 16, 16, 128, device=dev).to(K.BF16)
w = torch.randn(3, 3, 128, 128, device=dev) / 34.
(rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
xb = torch.randn(320, 32, 32, 256, device=dev).to(K.BF16)
wb = torch.randn(3, 3, 256, 256, device=dev) / 48.
wf, _ = K.prep_weights(wb, True, False)
g = torch.cuda.CUDAGraph()
def body():
    for _ in range(burst): K.conv2d_fprop(xb, wf, None, (32, 32), 256, 3)
    for _ in range(6): K.img16_conv3x3(x, rf, None, 128, K.IN_RELU)
body(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    body()
for _ in range(10): g.replay()
torch.cuda.synchronize()
