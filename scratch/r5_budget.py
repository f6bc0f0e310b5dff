"""Per-kernel conv FLOPs of one training iteration (counted by the launchers, one eagerly executed iteration with an event pair per
launch) and the time budget they imply at the rates this build's OWN best kernels sustain -- the replacement of DESIGN.md's
"ceiling ~0.30" estimate.  Classes: L = plain / phase 3x3 layers on 16x16 and 32x32 images with 256 channels (two-group, image-resident,
all-taps kernels), S = everything else (8x8 and 4x4 images, ConvMeanPool forms, 1x1, narrow-channel layers).
usage: python scratch/r5_budget.py  -> profiles/r05_phase_budget.txt (via gpurun_out/)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from gan_lib_tensorflow_amd import kernels as K
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
tr = S.SNGANTrainer(batch_size=64, seed=0, use_graphs=False)
feed = S.synthetic_batches(64, "cuda", seed=0)
for _ in range(3):
    tr.train_iteration(feed)
torch.cuda.synchronize()
K.prof_reset(); K.prof_enable(True)
tr.train_iteration(feed)
torch.cuda.synchronize()
K.prof_enable(False)
pair_us = 1e3 * K.prof_calibrate(200)
rows = []
for f, fam in ((0, "fprop/dgrad"), (1, "wgrad")):
    for nm, n, ms, fl, by in K.prof_kernels(f):
        net = max(ms - n * pair_us * 1e-3, 1e-6)
        rows.append((fam, nm, n, ms, net, fl))
L_KEYS = ("conv_igemm_pp_kernel", "img16_conv3x3", "conv_wgrad_taps_kernel<0", "conv_wgrad_taps_kernel<1", "conv_wgrad_rows_kernel<0, 2, true>")
tot_fl = sum(r[5] for r in rows)
print(f"event-pair floor {pair_us:.2f} us per launch (subtracted below); conv FLOPs as run {tot_fl / 1e9:.1f} GFLOP per iteration")
print(f"{'family':12s} {'kernel':70s} {'n':>3s} {'GFLOP':>8s} {'net ms':>8s} {'TFLOP/s':>8s} class")
best = {}
for fam, nm, n, ms, net, fl in sorted(rows, key=lambda r: -r[4]):
    cls = "L" if nm.startswith(L_KEYS) else "S"
    rate = fl / net / 1e9
    key = (fam, cls)
    if fl > 20e9:
        best[key] = max(best.get(key, 0.0), rate)
    print(f"{fam:12s} {nm[:70]:70s} {n:3d} {fl / 1e9:8.1f} {net:8.3f} {rate:8.0f} {cls}")
print()
budget = 0.0
for key in sorted(best):
    fl = sum(r[5] for r in rows if (r[0], "L" if r[1].startswith(L_KEYS) else "S") == key)
    ms = sum(r[4] for r in rows if (r[0], "L" if r[1].startswith(L_KEYS) else "S") == key)
    b = fl / best[key] / 1e9
    budget += b
    print(f"class {key}: {fl / 1e9:8.1f} GFLOP in {ms:6.3f} ms (net of the event floor); best kernel of the class {best[key]:5.0f} TFLOP/s -> budget {b:6.3f} ms")
print(f"conv budget at the best-of-class rates: {budget:.3f} ms per iteration; measured conv time (net) {sum(r[4] for r in rows):.3f} ms")
