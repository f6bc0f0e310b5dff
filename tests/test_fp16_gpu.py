"""fp16 operand path: libgank_f16.so is the same kernel sources built for IEEE-half buffers (v_mfma_f32_32x32x16_f16, fp32
accumulation).  The element type is a per-process choice (GANK_DTYPE), so the checks run in a child process
(tests/fp16_worker.py): every conv kernel family, conditional batch norm, and the SNGAN networks / losses / gradients against the
float64 oracle at half-precision tolerances, plus captured training iterations."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fp16_build_runs_the_path_against_the_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, GANK_DTYPE="fp16")
    env.pop("GANK_LIB_NAME", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fp16_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "FP16 PATH OK" in r.stdout and r.stdout.count("ok conv") == 5


def test_both_builds_export_the_same_abi():
    """CPU: libgank.so (bf16) and libgank_f16.so (fp16) are built from the same sources and export the same symbols; each
    reports its element type."""
    import ctypes
    from gan_lib_tensorflow_amd import _lib, build
    build.build(verbose=False)
    here = os.path.dirname(_lib.LIB_PATH)
    a, b = ctypes.CDLL(os.path.join(here, "libgank.so")), ctypes.CDLL(os.path.join(here, "libgank_f16.so"))
    assert a.gank_act_dtype() == 0 and b.gank_act_dtype() == 1 and a.gank_version() == b.gank_version()
    for name in _lib.PROTOTYPES:
        assert hasattr(a, name) and hasattr(b, name), name
