"""Would the LDS-patch phase kernel pay on G.Block.2.Conv1 (8x8 -> 16x16, 256 -> 256)?  The kernel needs 8 x 16 low-resolution patches;
two 8x8 images side by side have the pixel count and the per-block work of one 8x16 image, so time the kernel on [n/2, 8, 16, 256]
beside the generic kernel on [n, 8, 8, 256] (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) / 48.
wup = K.upconv3x3_prep(w)
b = torch.zeros(256, device=dev)
for n in (128, 320):
    xa = torch.randn(n, 8, 8, 256, device=dev).to(K.BF16)
    xb = torch.randn(n // 2, 8, 16, 256, device=dev).to(K.BF16)
    for _ in range(20):
        K.upconv3x3_fprop(xa, wup[0], b, 256, 0, None, 2)
    for _ in range(20):
        K.upconv3x3_fprop(xb, wup[0], b, 256, 0, None, 2)
    for _ in range(20):
        K.upconv3x3_fprop(xb, wup[0], b, 256, 0, None, 0)
torch.cuda.synchronize()
