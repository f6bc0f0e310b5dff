"""warm timing of the critic update's small filter gradients (128 samples): the batched 8x8x128 3x3 layers (filter-row kernel),
the 1x1 shortcut of D.Block.2 at 8x8 (lean kernel), the two ConvMeanPool 3x3 layers (filter-row stride-2 form + fold), D.Block.2.Conv1
(all-taps + slab reduce), the streaming 3-channel pair.  A/B two builds with GANK_LIB_NAME=<other .so>; a -DGANK_TUNING build reads the
dispatcher knobs (GANK_WGRAD_ROWS_TARGET, GANK_WGRAD_ROWS_PF, ...) from the environment."""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
dev = torch.device('cuda')
torch.manual_seed(0)
which = sys.argv[1:] or ['rows', 'lean', 'cpool', 'taps', 'narrow']
def warm(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
def rnd(*shape):
    return torch.randn(*shape, device=dev).to(K.BF16)
n = int(os.environ.get('N', 128))
if 'rows' in which:
    items = [(rnd(n, 8, 8, 128), rnd(n, 8, 8, 128), torch.zeros(3, 3, 128, 128, device=dev), torch.zeros(128, device=dev)) for _ in range(4)]
    t = warm(lambda: K.conv2d_wgrad_batched(items, (8, 8), 3, K.IN_RELU))
    print(f'batched 4 x (8x8, 128->128, 3x3, relu) n={n}: {t:7.1f} us ({4*n*64*128*1152*2/t/1e6:6.0f} TF)', flush=True)
    one = items[0]
    t = warm(lambda: K.conv2d_wgrad(one[0], one[1], one[2], (8, 8), 3, K.IN_RELU))
    print(f'single (8x8, 128->128, 3x3, relu) n={n}: {t:7.1f} us', flush=True)
    x, dy, dw = rnd(n, 8, 8, 256), rnd(n, 8, 8, 256), torch.zeros(3, 3, 256, 256, device=dev)
    t = warm(lambda: K.conv2d_wgrad(x, dy, dw, (8, 8), 3, 0))
    print(f'G.Block.1.Conv2 (8x8, 256->256, 3x3) n={n}: {t:7.1f} us ({n*64*256*2304*2/t/1e6:6.0f} TF)', flush=True)
if 'lean' in which:
    for (h, cin, cout, nn) in ((8, 256, 128, n), (4, 1024, 256, n), (8, 256, 256, n), (16, 256, 256, n)):
        x, dy, dw = rnd(nn, h, h, cin), rnd(nn, h, h, cout), torch.zeros(1, 1, cin, cout, device=dev)
        t = warm(lambda: K.conv2d_wgrad(x, dy, dw, (h, h), 1, 0))
        print(f'1x1 {h}x{h} {cin}->{cout} n={nn}: {t:7.1f} us ({nn*h*h*cin*cout*2/t/1e6:6.0f} TF)', flush=True)
if 'cpool' in which:
    for (hp, cin, cout) in ((16, 128, 128), (8, 256, 128)):
        x, dy, dw = rnd(n, 2 * hp, 2 * hp, cin), rnd(n, hp, hp, cout), torch.zeros(3, 3, cin, cout, device=dev)
        db = torch.zeros(cout, device=dev)
        t = warm(lambda: K.convpool3x3_wgrad(x, dy, dw, K.IN_RELU, db))
        print(f'ConvMeanPool 3x3 {2*hp}->{hp} {cin}->{cout} n={n}: {t:7.1f} us ({n*hp*hp*16*cin*cout*2/t/1e6:6.0f} TF as run)', flush=True)
if 'taps' in which:
    x, dy, dw = rnd(n, 16, 16, 256), rnd(n, 16, 16, 256), torch.zeros(3, 3, 256, 256, device=dev)
    t = warm(lambda: K.conv2d_wgrad(x, dy, dw, (16, 16), 3, K.IN_RELU))
    print(f'D.Block.2.Conv1 16x16 256->256 n={n}: {t:7.1f} us ({n*256*256*2304*2/t/1e6:6.0f} TF)', flush=True)
if 'narrow' in which:
    x0, dy0, dw0, db0 = rnd(n, 32, 32, 3), rnd(n, 32, 32, 128), torch.zeros(3, 3, 3, 128, device=dev), torch.zeros(128, device=dev)
    x1, dy1, dw1, db1 = rnd(n, 16, 16, 3), rnd(n, 16, 16, 128), torch.zeros(1, 1, 3, 128, device=dev), torch.zeros(128, device=dev)
    t = warm(lambda: K.conv2d_wgrad_narrow_pair((x0, dy0, dw0, db0, (32, 32), 3), (x1, dy1, dw1, db1, (16, 16), 1)))
    print(f'narrow pair (3->128 3x3 at 32x32 + 1x1 at 16x16) n={n}: {t:7.1f} us ({n*1024*128*2/t/1e3:6.0f} GB/s of dy)', flush=True)
