"""A synthetic class-conditional CIFAR-shaped dataset (10 classes, each a distinct low-frequency colour pattern + per-sample noise
and brightness) in the reference's row format (uint8 [N, 3072] CHW-planar, int32 labels: common/data/cifar10.py:9-15).  The only
sample-quality proxy available without Inception weights or CIFAR-10: used by tests/test_model_gpu.py and scratch/long_run.py."""
import numpy as np


def class_patterns():
    """[10, 3, 32, 32] float in [0, 1]: class c = a colour + a 2-D sinusoid with class-specific frequency and phase"""
    yy, xx = np.meshgrid(np.arange(32) / 32.0, np.arange(32) / 32.0, indexing='ij')
    pats = np.zeros((10, 3, 32, 32))
    for c in range(10):
        fy, fx = 1 + c % 3, 1 + (c // 3) % 3
        for ch in range(3):
            base = 0.25 + 0.5 * (((c * 7 + ch * 3) % 10) / 9.0)
            pats[c, ch] = base + 0.2 * np.sin(2 * np.pi * (fy * yy + fx * xx) + 0.7 * c + 2.1 * ch)
    return np.clip(pats, 0, 1)


def make_dataset(n_per_class, seed):
    rng = np.random.default_rng(seed)
    pats = class_patterns()
    labels = np.repeat(np.arange(10), n_per_class)
    x = pats[labels] * rng.uniform(0.85, 1.15, size=(len(labels), 1, 1, 1)) + rng.normal(0, 0.06, size=(len(labels), 3, 32, 32))
    u8 = np.clip(np.round(x * 255), 0, 255).astype(np.uint8).reshape(len(labels), 3072)          # CHW-planar rows (cifar10.py:9-15)
    perm = rng.permutation(len(labels))
    return u8[perm], labels[perm].astype(np.int32)
