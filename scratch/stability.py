"""Longer captured-graph training run on the synthetic feed: losses stay finite and bounded, parameters finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
tr = S.SNGANTrainer(batch_size=64, seed=0)
feed = S.synthetic_batches(64, torch.device("cuda"), seed=0)
hist = []
for it in range(int(os.environ.get("ITERS", "400"))):
    tr.train_iteration(feed)
    if it % 50 == 49:
        torch.cuda.synchronize()
        hist.append((it + 1, float(tr.d_loss), float(tr.g_loss)))
        print(hist[-1], flush=True)
ok = all(torch.isfinite(v).all() for v in tr.store.vars.values())
print("all parameters finite:", ok)
sys.exit(0 if ok and all(abs(d) < 10 and abs(g) < 10 for _, d, g in hist) else 1)
