"""run-to-run difference of the generator gradient buffer (same seeds, same RNG state): what fp32-atomics ordering alone does"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
for bs in (8, 64):
    tr = S.SNGANTrainer(batch_size=bs, seed=17, use_graphs=False)
    rng0 = tr.rng_state.clone()
    outs = []
    for _ in range(3):
        tr.rng_state.copy_(rng0)
        tr._g_forward_backward()
        torch.cuda.synchronize()
        outs.append(tr.g_flat["grads"].clone())
    print("batch", bs, "rel diff run0-run1", float((outs[0] - outs[1]).norm() / outs[0].norm()), "run0-run2", float((outs[0] - outs[2]).norm() / outs[0].norm()), flush=True)
    del tr
