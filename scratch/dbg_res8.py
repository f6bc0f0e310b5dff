import sys, os
sys.path.insert(0, '.')
import torch
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
from gan_lib_tensorflow_amd import functional as Fn, kernels as K
tr = S.SNGANTrainer(batch_size=64, seed=0, use_graphs=False)
for k, v in tr._g_convs:
    print(k, tuple(v.shape), S._g_prep_kind(k, v), [a for a in ('_prep', '_prep_up', '_prep_res') if getattr(v, a, None) is not None])
orig = K.res8_conv3x3
def spy(x, *a, **kw):
    print('res8_conv3x3 called', tuple(x.shape), a[3] if len(a) > 3 else None)
    return orig(x, *a, **kw)
K.res8_conv3x3 = spy
orig2 = K.upconv3x3_fprop
def spy2(x, *a, **kw):
    print('upconv3x3_fprop called', tuple(x.shape))
    return orig2(x, *a, **kw)
K.upconv3x3_fprop = spy2
tr._g_forward_backward()
torch.cuda.synchronize()
