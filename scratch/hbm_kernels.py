"""Achieved HBM rate of the bandwidth-bound kernels at their largest shape in the step (algorithmic bytes / rocprofv3-style
event time over 30 launches).  Peak: 8 TB/s (MI355X_MICROARCH.md)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(3)
def bfr(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
def timeit(run, reps=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
rows = []
def rec(name, shape, nbytes, run):
    us = timeit(run)
    rows.append((name, shape, nbytes / 1e6, us, nbytes / us / 1e3))
# CBN at the generator's last site, critic-feed pass (n=320) and update pass (n=128)
for n, groups in ((320, 10), (128, 2)):
    x = bfr(n, 32, 32, 256); labels = torch.randint(0, 10, (n,), device="cuda", dtype=torch.int32)
    gamma = torch.randn(10, 256, device="cuda"); beta = torch.randn(10, 256, device="cuda")
    eb = x.numel() * 2
    rec("cbn_fwd (stats + finalize + apply, relu)", f"[{n},32,32,256]", 3 * eb, lambda: K.cbn_fwd(x, labels, gamma, beta, groups, True))
    y, stats = K.cbn_fwd(x, labels, gamma, beta, groups, True)[:2]
    dy = bfr(n, 32, 32, 256); dg = torch.zeros(10, 256, device="cuda"); db = torch.zeros(10, 256, device="cuda")
    rec("cbn_bwd (sums + tables + apply)", f"[{n},32,32,256]", 7 * eb, lambda: K.cbn_bwd(dy, x, y, labels, gamma, stats, dg, db, groups, True))
x = bfr(128, 32, 32, 128)
rec("pool2x2", "[128,32,32,128]", x.numel() * 2 * 1.25, lambda: K.pool2x2(x))
g = bfr(128, 16, 16, 128)
rec("unpool2x2_add", "[128,16,16,128]", g.numel() * 2 * 5, lambda: K.unpool2x2_add(g, None, 0.25))
a = bfr(128, 8, 8, 128); b = bfr(128, 8, 8, 128)
rec("add", "[128,8,8,128]", a.numel() * 2 * 3, lambda: K.add(a, b))
nG = 7875587 // 4 * 4
p = torch.randn(nG, device="cuda"); gr = torch.randn(nG, device="cuda"); m = torch.zeros(nG, device="cuda"); v = torch.zeros(nG, device="cuda")
hp = torch.tensor([2e-4, 0.0, 0.9, 1e-8, 1.0, 0.0, 0.0, 0.0], device="cuda"); t = torch.zeros(1, dtype=torch.int64, device="cuda")
rec("adam_tf (generator, 7.9 M parameters)", f"[{nG}]", nG * 28, lambda: K.adam_tf(p, gr, m, v, hp, t))
x3 = bfr(128, 32, 32, 3); w3 = torch.randn(3, 3, 3, 128, device="cuda") / 27 ** 0.5
wf3, _ = K.prep_weights(w3, True, False)
rec("conv_narrow_in 3x3 3->128 (output-write bound)", "[128,32,32,3]->[.,128]", 128 * 1024 * (3 + 128) * 2, lambda: K.conv2d_fprop(x3, wf3, None, (32, 32), 128, 3))
print(f"{'kernel':50s} {'shape':26s} {'MB':>8s} {'us':>8s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for name, shape, mb, us, gbs in rows:
    print(f"{name:50s} {shape:26s} {mb:8.1f} {us:8.1f} {gbs:8.0f} {gbs / 8000:9.2f}")
