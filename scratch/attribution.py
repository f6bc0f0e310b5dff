"""Which stored tensor carries the generator-gradient error of a bf16 implementation?  The float64 restatement with every
stored tensor rounded to bf16 (oracle/ref_torch.STORE) against exact arithmetic, then with ONE class of stored tensors (and
one direction) left exact at a time.  CPU only; writes the table DESIGN.md section 2 quotes.
usage: python scratch/attribution.py [samples_per_tower=32]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import ref_torch as T  # noqa: E402

torch.set_num_threads(8)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
state = T.init_sngan_params(21)
rng = np.random.default_rng(64)
z = torch.tensor(rng.normal(size=(2 * b, 128))).to(torch.bfloat16).to(torch.float64)
fl = torch.tensor(rng.integers(0, 10, 2 * b))
names = ['Generator/G.Output/Filters', 'Generator/G.Block.3.Conv2/Filters', 'Generator/G.Block.3.Conv1/Filters', 'Generator/G.Block.2.Conv2/Filters',
         'Generator/G.Block.2.Conv1/Filters', 'Generator/G.Block.1.Conv2/Filters', 'Generator/G.Block.1.Conv1/Filters', 'Generator/G.Input/W']


def grads(store, exact=()):
    T.STORE, T.STORE_EXACT = store, frozenset(exact)
    try:
        P = T.to_torch(state)
        loss, _ = T.g_loss_fn(P, z, fl)
        return dict(zip(names, torch.autograd.grad(loss, [P[k] for k in names])))
    finally:
        T.STORE, T.STORE_EXACT = None, frozenset()


g0 = grads(None)
rows = [("every stored tensor rounded (both directions)", T.bf16_storage, ()),
        ("values rounded, gradients exact", T.bf16_storage_fwd, ()),
        ("gradients rounded, values exact", T.bf16_storage_bwd, ())]
for cls in ('cbn', 'conv1', 'block', 'short', 'lin', 'image', 'd'):
    rows.append((f"all rounded except '{cls}'", T.bf16_storage, (cls,)))
rows.append(("only the critic's tensors rounded", T.bf16_storage, ('cbn', 'conv1', 'block', 'short', 'lin', 'image')))
rows.append(("only 'cbn' + 'conv1' rounded (values)", T.bf16_storage_fwd, ('block', 'short', 'lin', 'image', 'd')))
print(f"relative L2 error of generator filter gradients vs exact float64, {2 * b} samples (2 towers of {b})")
print(f"{'':46s}" + "".join(f"{n.split('/')[1].replace('G.', '').replace('Block.', 'B'):>10s}" for n in names))
for label, store, exact in rows:
    g = grads(store, exact)
    print(f"{label:46s}" + "".join(f"{float((g[k] - g0[k]).norm() / g0[k].norm()):10.4f}" for k in names), flush=True)
