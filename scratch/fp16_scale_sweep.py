"""fp16 build: generator-gradient error against the float64 oracle at the headline batch for several static loss scales.
run: GANK_DTYPE=fp16 python scratch/fp16_scale_sweep.py"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from gan_lib_tensorflow_amd import kernels as K
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
from oracle import ref_torch as T
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
b = 64
gref = None
for scale in (64.0, 256.0, 1024.0, 4096.0, 16384.0, 65536.0):
    tr = S.SNGANTrainer(batch_size=b, seed=21, use_graphs=False, loss_scale=scale)
    rng2 = np.random.default_rng(64)
    z2 = torch.tensor(rng2.normal(size=(2 * b, 128)).astype(np.float32)).to(K.BF16)
    fl = torch.tensor(rng2.integers(0, 10, 2 * b), dtype=torch.int32)
    if gref is None:
        P = T.to_torch(tr.store.state_dict())
        loss_ref, _ = T.g_loss_fn(P, z2.double(), fl.long())
        gn = [k for k in T.trainable_names(P, 'Generator') if k.endswith(('Filters', '/W'))]
        gref = dict(zip(gn, torch.autograd.grad(loss_ref, [P[k] for k in gn])))
    errs = []
    for rep in range(2):
        tr.g_flat['grads_all'].zero_(); tr.g_flat['clean'] = True
        tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
        torch.cuda.synchronize()
        e = {}
        for k in gn:
            g, r = tr.store.vars[k].main_grad.double().cpu().flatten() / scale, gref[k].flatten()
            e[k] = float((g - r).norm() / r.norm())
        errs.append(e)
    tr.g_opt.apply(); torch.cuda.synchronize()
    print(f"scale {scale:8.0f}: G.Input {errs[0]['Generator/G.Input/W']:.4f}/{errs[1]['Generator/G.Input/W']:.4f}  B1.Conv1 {errs[0]['Generator/G.Block.1.Conv1/Filters']:.4f}  "
          f"B2.Conv1 {errs[0]['Generator/G.Block.2.Conv1/Filters']:.4f}  B3.Conv1 {errs[0]['Generator/G.Block.3.Conv1/Filters']:.4f}  Output {errs[0]['Generator/G.Output/Filters']:.4f}  health {tr.health()}", flush=True)
    del tr
