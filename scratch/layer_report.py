"""Per-layer conv report of one eager training iteration: (kind, shape) -> launches, time, TFLOP/s.
Times are torch events around each wrapper call (they include the ~3 us eager launch gap)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd import kernels as K
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S

recs = []


def wrap(name, flops_fn):
    orig = getattr(K, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        recs.append((name, flops_fn(*a, **k), e0, e1))
        return out
    setattr(K, name, f)


def fl_fprop(x, wf, bias, out_hw, cout, ksize, flags=0, *a, **k):
    n, cin = x.shape[0], x.shape[3]
    return (f"N{n} {out_hw[0]}x{out_hw[1]} k{ksize} {cin}->{cout} f{flags}", 2.0 * n * out_hw[0] * out_hw[1] * ksize * ksize * cin * cout)


def fl_dgrad(dy, wd, out_hw, cin, ksize, flags=0, *a, **k):
    n, cout = dy.shape[0], dy.shape[3]
    return (f"N{n} {out_hw[0]}x{out_hw[1]} k{ksize} {cin}<-{cout} f{flags}", 2.0 * n * out_hw[0] * out_hw[1] * ksize * ksize * cin * cout)


def fl_wgrad(x, dy, dw, hw, ksize, flags=0, *a, **k):
    n, cin, cout = x.shape[0], x.shape[3], dy.shape[3]
    return (f"N{n} {hw[0]}x{hw[1]} k{ksize} {cin}x{cout} f{flags}", 2.0 * n * hw[0] * hw[1] * ksize * ksize * cin * cout)


def fl_upf(x, wph, bias, cout, *a, **k):
    n, hl, wl, cin = x.shape
    return (f"N{n} {2*hl}x{2*wl} up3x3 {cin}->{cout}", 2.0 * n * 4 * hl * wl * 4 * cin * cout)


def fl_upd(dy, wd4, cin, *a, **k):
    n, h2, w2, cout = dy.shape
    return (f"N{n} {h2}x{w2} up3x3 {cin}<-{cout}", 2.0 * n * (h2 // 2) * (w2 // 2) * 16 * cin * cout)


def fl_cpf(x, wp4, bias, cout, *a, **k):
    n, h, w, cin = x.shape
    return (f"N{n} {h}x{w} cpool3x3 {cin}->{cout}", 2.0 * n * (h // 2) * (w // 2) * 16 * cin * cout)


def fl_cpd(dy, wphd, cin, *a, **k):
    n, hp, wp, cout = dy.shape
    return (f"N{n} {2*hp}x{2*wp} cpool3x3 {cin}<-{cout}", 2.0 * n * hp * wp * 16 * cin * cout)


def fl_cpw(x, dy, dw, *a, **k):
    n, hp, wp, cout = dy.shape
    return (f"N{n} {2*hp}x{2*wp} cpool3x3 {x.shape[3]}x{cout}", 2.0 * n * hp * wp * 16 * x.shape[3] * cout)


for nm, fn in (("convpool3x3_fprop", fl_cpf), ("convpool3x3_dgrad", fl_cpd), ("convpool3x3_wgrad", fl_cpw), ("conv2d_fprop", fl_fprop), ("conv2d_dgrad", fl_dgrad), ("conv2d_wgrad", fl_wgrad),
               ("upconv3x3_fprop", fl_upf), ("upconv3x3_dgrad", fl_upd)):
    wrap(nm, fn)

tr = S.SNGANTrainer(batch_size=64, device="cuda", seed=0, use_graphs=False)
feed = S.synthetic_batches(64, tr.device, seed=0)
for _ in range(3):
    tr.train_iteration(feed)
torch.cuda.synchronize()
recs.clear()
tr.train_iteration(feed)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, (shape, fl), e0, e1 in recs:
    k = (name, shape)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"total conv time {tot:.3f} ms over {len(recs)} launches")
for (name, shape), (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:16s} {shape:40s} x{c:3d} {ms:7.3f} ms  {1e3*ms/c:7.1f} us/launch {fl/ms/1e9:8.1f} TF/s  {100*ms/tot:5.1f}%")
