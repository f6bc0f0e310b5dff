"""Bit-identity screen over the atomics-free kernels at the network's shapes (fprop / dgrad / phase forms / CBN fwd / SN fwd)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lib_tensorflow_amd import kernels as K
torch.manual_seed(2)
REPS = int(os.environ.get("REPS", "200"))
side = torch.cuda.Stream(); junk = torch.randn(32 << 20, device="cuda")
def screen(name, run):
    ref = run()
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
    torch.cuda.synchronize()
    d = 0
    for rep in range(REPS):
        if rep % 3 == 0:
            with torch.cuda.stream(side): junk.mul_(1.0001)
        y = run(); y = y if isinstance(y, (tuple, list)) else [y]
        d += 0 if all(torch.equal(a, b) for a, b in zip(y, ref)) else 1
    torch.cuda.synchronize()
    print(f"{name:48s} differing {d}/{REPS}", flush=True)
    return d
bad = 0
def bfr(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
for (n, h, cin, cout, k, relu) in [(128, 8, 128, 128, 3, 1), (128, 16, 256, 256, 3, 1), (128, 32, 256, 256, 3, 0), (320, 32, 256, 3, 3, 0), (128, 32, 3, 128, 3, 0),
                                   (128, 16, 3, 128, 1, 0), (128, 8, 256, 128, 1, 0), (320, 8, 256, 256, 3, 0), (128, 16, 256, 256, 1, 0)]:
    x = bfr(n, h, h, cin); w = torch.randn(k, k, cin, cout, device="cuda") / (k * k * cin) ** 0.5
    wf, wd = K.prep_weights(w, True, True)
    b = torch.randn(cout, device="cuda")
    bad += screen(f"fprop n{n} {h}x{h} k{k} {cin}->{cout} relu{relu}", lambda: K.conv2d_fprop(x, wf, b, (h, h), cout, k, K.IN_RELU if relu else 0))
    dy = bfr(n, h, h, cout)
    bad += screen(f"dgrad n{n} {h}x{h} k{k} {cin}<-{cout}", lambda: K.conv2d_dgrad(dy, wd, (h, h), cin, k, 0, 1.0, None, x if relu else None))
for (n, hl, cin, cout) in [(320, 4, 1024, 256), (128, 8, 256, 256), (128, 16, 256, 256)]:
    x = bfr(n, hl, hl, cin); w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
    wph, wd4 = K.upconv3x3_prep(w)
    bad += screen(f"upconv fprop n{n} {hl}x{hl} {cin}->{cout}", lambda: K.upconv3x3_fprop(x, wph, None, cout))
    dy = bfr(n, 2 * hl, 2 * hl, cout)
    bad += screen(f"upconv dgrad n{n} {hl}x{hl}", lambda: K.upconv3x3_dgrad(dy, wd4, cin))
for (n, hp, cin, cout) in [(128, 16, 128, 128), (128, 8, 256, 128)]:
    x = bfr(n, 2 * hp, 2 * hp, cin); w = torch.randn(3, 3, cin, cout, device="cuda") / (9 * cin) ** 0.5
    wp4, wphd = K.convpool3x3_prep(w)
    bad += screen(f"convpool fprop n{n} {hp}x{hp} {cin}->{cout}", lambda: K.convpool3x3_fprop(x, wp4, None, cout, K.IN_RELU))
    dy = bfr(n, hp, hp, cout)
    bad += screen(f"convpool dgrad n{n} {hp}x{hp}", lambda: K.convpool3x3_dgrad(dy, wphd, cin, x))
# CBN forward (two-pass statistics, deterministic merge)
for (n, h, c, groups) in [(320, 32, 256, 10), (128, 8, 256, 2), (128, 4, 1024, 2)]:
    x = bfr(n, h, h, c); labels = torch.randint(0, 10, (n,), device="cuda", dtype=torch.int32)
    gamma = torch.randn(10, c, device="cuda"); beta = torch.randn(10, c, device="cuda")
    bad += screen(f"cbn fwd n{n} {h}x{h} c{c}", lambda: K.cbn_fwd(x, labels, gamma, beta, groups, True)[0])
print("TOTAL", bad)
sys.exit(1 if bad else 0)
