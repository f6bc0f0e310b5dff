import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(0)
for (n, cin, cout) in ((1, 128, 128), (2, 128, 256), (2, 256, 128), (2, 256, 256)):
    x = torch.randn(n, 16, 16, cin, device='cuda').to(K.BF16)
    w = torch.randn(3, 3, cin, cout, device='cuda') / (3 * cin ** 0.5)
    (rf, rd), = K.prep_weights_batched([w], want_d=True, kinds=[4])
    wf, wd = K.prep_weights(w, True, True)
    y = K.img16_conv3x3(x, rf, None, cout, 0)
    y0 = K.conv2d_fprop(x, wf, None, (16, 16), cout, 3, 0)
    dy = torch.randn(n, 16, 16, cout, device='cuda').to(K.BF16)
    dx = K.img16_conv3x3(dy, rd, None, cin, 0)
    dx0 = K.conv2d_dgrad(dy, wd, (16, 16), cin, 3)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(dx.float())
    print(n, cin, cout, "fprop max diff", float((y.float() - y0.float()).abs().max()), "dgrad nan", int(bad.sum()), "max diff", float((dx.float() - dx0.float())[~bad].abs().max()))
    if bad.any():
        idx = bad.nonzero()
        print("   first bad", idx[:4].tolist(), "channels", sorted(set(idx[:, 3].tolist()))[:10], "rows", sorted(set(idx[:, 1].tolist())))
