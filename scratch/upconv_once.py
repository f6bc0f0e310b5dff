"""a few launches of the 16x16 -> 32x32 UpsampleConv (two-group kernel, phase form) and of the plain 32x32 conv, for PMC passes"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
w = torch.randn(3, 3, 256, 256, device=dev) * 0.02
wph, _ = K.upconv3x3_prep(w)
wf = K.prep_weights(w)[0]
bias = torch.zeros(256, device=dev)
x16 = torch.randn(128, 16, 16, 256, device=dev).to(K.BF16)
x32 = torch.randn(128, 32, 32, 256, device=dev).to(K.BF16)
for sg in (0, 2):
    for _ in range(3):
        K.upconv3x3_fprop(x16, wph, bias, 256, stats_groups=sg)
        torch.cuda.synchronize()
if wf is not None:
    for sg in (0, 2):
        for _ in range(3):
            K.conv2d_fprop(x32, wf, bias, (32, 32), 256, 3, stats_groups=sg)
            torch.cuda.synchronize()
