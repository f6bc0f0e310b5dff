"""CPU oracle for the SNGAN-ResNet CIFAR-10 hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (watsonyanghx/GAN_Lib_Tensorflow) ships no tests, golden
vectors or fixtures for this path, its arithmetic lives in TensorFlow 1.5 (pip dependency,
README.md:11, not vendored) and TensorFlow is not importable in this pipeline
(ModuleNotFoundError at common/__init__.py:3 -- an ordinary Python error, not a denial).
The oracle is therefore a restatement pinned by (1) two independent implementations that
must agree (NumPy float64 with hand-derived gradients in `ref_ops.py`, torch-CPU float64
autograd in `ref_torch.py`), (2) known-answer tests, (3) finite differences.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package.  The product package `gan_lib_tensorflow_amd` never does.
"""
