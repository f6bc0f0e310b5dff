import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_torch as T
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S

def mk(graphs, seed=11, b=8):
    return S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=graphs, state=T.init_sngan_params(seed))

def cmp(tag, a, b):
    d = (a - b).abs()
    print(f"{tag}: max {d.max().item():.3e} mean {d.mean().item():.3e} ref max {a.abs().max().item():.3e} frac>1e-6*max {(d > 1e-6 * a.abs().max()).float().mean().item():.3f}")

for mode in ("eager-vs-eager", "eager-vs-graph"):
    print("=====", mode)
    ta, tb = mk(False), mk(mode == "eager-vs-graph")
    fa, fb = S.synthetic_batches(8, "cuda", seed=1), S.synthetic_batches(8, "cuda", seed=1)
    for step in range(4):
        da, la = next(fa); db, lb = next(fb)
        ta.d_step(da, la); tb.d_step(db, lb)
        torch.cuda.synchronize()
        print("step", step, "loss", float(ta.d_loss), float(tb.d_loss), "rng", ta.rng_state.tolist(), tb.rng_state.tolist())
        cmp("  D grads ", ta.d_flat["grads"], tb.d_flat["grads"])
        cmp("  D params", ta.d_flat["params"], tb.d_flat["params"])
        cmp("  u       ", ta.store.vars['Discriminator/D.Block.2.Conv1/filters/spectral_norm/u'], tb.store.vars['Discriminator/D.Block.2.Conv1/filters/spectral_norm/u'])
    ta.g_step(); tb.g_step(); torch.cuda.synchronize()
    print("g loss", float(ta.g_loss), float(tb.g_loss))
    cmp("  G grads ", ta.g_flat["grads"], tb.g_flat["grads"])
    ta.g_step(); tb.g_step(); torch.cuda.synchronize()
    cmp("  G grads2", ta.g_flat["grads"], tb.g_flat["grads"])
