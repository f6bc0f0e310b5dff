import torch, sys
sys.path.insert(0, '/root/repo')
from gan_lib_tensorflow_amd import functional as Fn, kernels as K
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
orig = K.sum_slabs
def spy(jobs, label=None):
    print("sum_slabs: jobs", sum(c for _, c, _ in jobs), "label", label is not None, "label_jobs pending", len(Fn._label_jobs))
    return orig(jobs, label)
K.sum_slabs = spy
tr = S.SNGANTrainer(batch_size=64, seed=0)
def batches():
    while True:
        yield torch.randint(0, 256, (64, 3072), dtype=torch.uint8, device='cuda'), torch.randint(0, 10, (64,), dtype=torch.int32, device='cuda')
b = batches()
tr.use_graphs = False if hasattr(tr, 'use_graphs') else None
tr.train_iteration(b)
torch.cuda.synchronize()
