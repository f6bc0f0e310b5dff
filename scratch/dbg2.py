import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_torch as T
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
torch.set_num_threads(16)
def bf16r(a): return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)
for b in (4, 32):
    seed = 5
    state = T.init_sngan_params(seed)
    tr = S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=False, state=state)
    rng = np.random.default_rng(7)
    P = T.to_torch(tr.store.state_dict())
    z2 = bf16r(rng.normal(size=(2 * b, 128)))
    fl = torch.tensor(rng.integers(0, 10, 2 * b), dtype=torch.int32)
    loss, _ = T.g_loss_fn(P, z2.to(torch.float64), fl.long())
    gn = T.trainable_names(P, 'Generator')
    ref_g = dict(zip(gn, torch.autograd.grad(loss, [P[k] for k in gn])))
    tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
    torch.cuda.synchronize()
    print("b", b, "loss", float(tr.g_loss), float(loss))
    for k in gn:
        g = tr.store.vars[k].main_grad.double().cpu().flatten(); r = ref_g[k].flatten()
        mx = (g - r).abs().max() / r.abs().max().clamp_min(1e-30)
        cos = (g @ r) / (g.norm() * r.norm()).clamp_min(1e-30)
        l2 = (g - r).norm() / r.norm().clamp_min(1e-30)
        if 'Biases' in k and 'Output' not in k: continue
        print(f"  {k.split('/',1)[1]:40s} maxrel {mx:.4f} l2rel {l2:.4f} cos {cos:.5f}")
