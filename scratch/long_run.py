"""Quality proxy without Inception weights or CIFAR: train the SNGAN step on a SYNTHETIC class-conditional dataset (10 classes,
each a distinct low-frequency colour pattern + per-sample noise and brightness), then compare what the generator learned with
the data: per-class mean image, per-class per-pixel standard deviation, critic / generator loss windows.  Run once per element
type (GANK_DTYPE=bf16 | fp16) and compare the two result files with scratch/long_run_compare.py.
usage: [GANK_DTYPE=fp16] python scratch/long_run.py <iterations> <out.npz>"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gan_lib_tensorflow_amd import kernels as K  # noqa: E402
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S  # noqa: E402


from tests.synthetic_classes import make_dataset  # noqa: E402


def main():
    iters, out = int(sys.argv[1]), sys.argv[2]
    data, labels = make_dataset(500, 0)
    d_dev, l_dev = torch.tensor(data).cuda(), torch.tensor(labels).cuda()
    tr = S.SNGANTrainer(batch_size=64, seed=0)
    g = torch.Generator(device='cpu').manual_seed(1)

    def batches():
        while True:
            idx = torch.randperm(len(labels), generator=g)[:64].cuda()
            yield d_dev[idx].contiguous(), l_dev[idx].contiguous()
    feed = batches()
    dl, gl = [], []
    t0 = time.time()
    for it in range(iters):
        tr.train_iteration(feed)
        if it % 25 == 24:
            dl.append(float(tr.d_loss)); gl.append(float(tr.g_loss))
        if it % 500 == 499:
            print(f"it {it + 1}: d_loss {dl[-1]:.3f} g_loss {gl[-1]:.3f}  ({time.time() - t0:.0f} s)  health {tr.health()}", flush=True)
    torch.cuda.synchronize()
    # what the generator learned: 1000 samples per class in batches of 100 (batch statistics, as the reference samples: :530-555)
    means, stds = np.zeros((10, 3072)), np.zeros((10, 3072))
    for c in range(10):
        xs = []
        for _ in range(10):
            lab = torch.full((100,), c, dtype=torch.int32, device='cuda')
            xs.append(tr.sample(100, labels=lab).float().cpu().numpy())           # HWC order, tanh range
        x = np.concatenate(xs)
        means[c], stds[c] = x.mean(0), x.std(0)
    # the data in the same representation: 2 (u/256 - .5), HWC
    real = (2.0 * (data.astype(np.float64) / 256.0 - 0.5)).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1).reshape(-1, 3072)
    rmeans = np.stack([real[labels == c].mean(0) for c in range(10)])
    rstds = np.stack([real[labels == c].std(0) for c in range(10)])
    finite = bool(torch.isfinite(tr.g_flat["params"]).all() and torch.isfinite(tr.d_flat["params"]).all())
    np.savez(out, d_loss=np.array(dl), g_loss=np.array(gl), means=means, stds=stds, rmeans=rmeans, rstds=rstds, finite=finite,
             dtype=os.environ.get("GANK_DTYPE", "bf16"), iters=iters, seconds=time.time() - t0)
    err = np.linalg.norm(means - rmeans, axis=1) / np.linalg.norm(rmeans, axis=1)
    print("per-class |mean_G - mean_data| / |mean_data|:", np.round(err, 3), " finite:", finite, flush=True)


if __name__ == "__main__":
    main()
