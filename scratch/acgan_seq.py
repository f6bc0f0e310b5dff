"""ACGAN at 32 samples per GPU under rocprofv3 --kernel-trace: a few captured steps; scratch/acgan_seq.sh prints the kernel
sequence of one step (launch count, time per kernel symbol, the sequence itself)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches
from gan_lib_tensorflow_amd.ACGAN.train import ACGANTrainer
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
feed = synthetic_batches(bs, "cuda", seed=2)
tr = ACGANTrainer(batch_size=bs, seed=1)
for it in range(1, 9):
    tr.train_iteration(feed, it)
torch.cuda.synchronize()
