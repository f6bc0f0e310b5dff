import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gan_lib_tensorflow_amd import kernels as K
g = torch.Generator(device="cpu").manual_seed(11)
def mk(s): return torch.randn(s, generator=g).cuda()
def step(name, ws, kinds):
    print("start", name, flush=True)
    K.prep_weights_batched(ws, want_d=True, kinds=kinds)
    torch.cuda.synchronize()
    print("ok", name, flush=True)
step("plain", [mk((3, 3, 64, 64))], [0])
step("lin", [mk((128, 96))], [0])
step("up", [mk((3, 3, 64, 128))], [1])
step("pool", [mk((3, 3, 128, 64))], [2])
step("mixed", [mk((3, 3, 64, 128)), mk((3, 3, 128, 64)), mk((3, 3, 64, 64)), mk((128, 96)), mk((1, 1, 64, 32))], [1, 2, 0, 0, None])
step("again-default-kinds", [mk((3, 3, 64, 64)), mk((128, 96))], None)
