"""Which cold operand costs the 16x16 image kernel its in-step time?  Run under rocprofv3 --kernel-trace --stats: the kernel after
(a) nothing, (b) a rewrite of its weights, (c) a rewrite of its input, (d) both, (e) a 1-GB fill that evicts L2 and Infinity Cache."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
mode = sys.argv[1]
n = 128
x = torch.randn(n, 16, 16, 128, device=dev).to(K.BF16); x2 = x.clone()
w = torch.randn(3, 3, 128, 128, device=dev) / 34.
(rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
rf2 = rf.clone()
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)
for _ in range(30):
    if mode in ("w", "wx"): rf.copy_(rf2)
    if mode in ("x", "wx"): x.copy_(x2)
    if mode == "evict": big.fill_(1.0)
    K.img16_conv3x3(x, rf, None, 128, K.IN_RELU)
torch.cuda.synchronize()
