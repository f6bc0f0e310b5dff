"""warm timing of the batched weight preparation launches (generator: after its Adam; critic: inside every update) per kind"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from gan_lib_tensorflow_amd import kernels as K
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as M
dev = torch.device('cuda')
def warm(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000
tr = M.SNGANTrainer(batch_size=64, device='cuda', use_graphs=False)
g = tr._g_convs
kinds = [M._g_prep_kind(k, v) for k, v in g]
print('G all:', round(warm(lambda: K.prep_weights_batched([v for _, v in g], want_d=True, kinds=kinds)), 1), 'us')
for kd in sorted(set(k for k in kinds if k is not None)):
    sel = [(k, v) for (k, v), kk in zip(g, kinds) if kk == kd]
    print(f'  G kind {kd}: {len(sel)} weights {sum(v.numel() for _, v in sel)/1e6:.2f} M  ',
          round(warm(lambda: K.prep_weights_batched([v for _, v in sel], want_d=True, kinds=[kd] * len(sel))), 1), 'us', [tuple(v.shape) for _, v in sel][:4])
d = [(k, v) for k, v in tr.store.vars.items() if k.startswith('Discriminator') and k.endswith('Filters')]
kinds = [M._d_prep_kind(k, v) for k, v in d]
dd = [(kv, kd) for kv, kd in zip(d, kinds) if kd is not None]
print('D all:', round(warm(lambda: K.prep_weights_batched([kv[1] for kv, _ in dd], want_d=True, kinds=[kd for _, kd in dd])), 1), 'us')
for kd in sorted(set(k for _, k in dd)):
    sel = [kv for kv, kk in dd if kk == kd]
    print(f'  D kind {kd}: {len(sel)} weights {sum(v.numel() for _, v in sel)/1e6:.2f} M  ',
          round(warm(lambda: K.prep_weights_batched([v for _, v in sel], want_d=True, kinds=[kd] * len(sel))), 1), 'us', [tuple(v.shape) for _, v in sel][:4])
