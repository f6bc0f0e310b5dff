"""decoder_3's phase-stacked conv at batch 16 (x [16,64,64,512] -> 512, 3x3): wgrad / dgrad / fprop finite? batch 16 = sum of four batch-4 parts?"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(0)
for (hw, cin, cout) in ((64, 512, 512), (32, 1024, 1024)):
    n = 16
    x = torch.randn(n, hw, hw, cin, device='cuda').to(K.BF16)
    dy = (torch.randn(n, hw, hw, cout, device='cuda') * 0.01).to(K.BF16)
    w = torch.randn(3, 3, cin, cout, device='cuda') / (3 * cin ** 0.5)
    wf, wd = K.prep_weights(w, True, True)
    for flags in (0, K.IN_RELU):
        dw = torch.zeros_like(w)
        K.conv2d_wgrad(x, dy, dw, (hw, hw), 3, flags)
        parts = torch.zeros_like(w)
        for i in range(4):
            K.conv2d_wgrad(x[4 * i:4 * i + 4].contiguous(), dy[4 * i:4 * i + 4].contiguous(), parts, (hw, hw), 3, flags)
        torch.cuda.synchronize()
        print(f"hw {hw} {cin}->{cout} flags {flags}: wgrad finite {bool(torch.isfinite(dw).all())}, parts finite {bool(torch.isfinite(parts).all())}, "
              f"rel diff {float((dw - parts).norm() / parts.norm()):.2e}, nan count {int((~torch.isfinite(dw)).sum())}")
    y = K.conv2d_fprop(x, wf, None, (hw, hw), cout, 3, K.IN_RELU)
    dx = K.conv2d_dgrad(dy, wd, (hw, hw), cin, 3, 0, 1.0, None, x)
    torch.cuda.synchronize()
    print("   fprop finite", bool(torch.isfinite(y.float()).all()), "dgrad finite", bool(torch.isfinite(dx.float()).all()))
