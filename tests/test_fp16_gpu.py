"""fp16 operand path: libgank_f16.so is the same kernel sources built for IEEE-half buffers (v_mfma_f32_32x32x16_f16, fp32
accumulation).  The element type is a per-process choice (GANK_DTYPE), so the checks run in ONE child process per session
(tests/fp16_worker.py) that records a result per section; every section is a test id of its own here: every conv kernel family,
conditional batch norm, the SNGAN networks / losses / gradients against the float64 oracle at half-precision tolerances, captured
training iterations, and the optimiser's non-finite-gradient guard."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SECTIONS = ["conv two-group", "conv patch", "conv generic 8x8", "conv 1x1", "conv narrow input", "upconv phase form",
            "wgrad all-taps", "cond_batchnorm", "SNGAN critic loss + gradients", "SNGAN generator loss + gradients",
            "SNGAN training iterations under hipGraph replay", "Adam skips non-finite gradients",
            "SNGAN batch-64 generator gradients, loss scale 1024"]


@pytest.fixture(scope="module")
def fp16_results(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = tmp_path_factory.mktemp("fp16") / "results.json"
    env = dict(os.environ, GANK_DTYPE="fp16")
    env.pop("GANK_LIB_NAME", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fp16_worker.py"), str(out)], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    results = json.loads(out.read_text()) if out.exists() else {}
    return {"results": results, "rc": r.returncode, "tail": r.stdout[-2000:] + r.stderr[-3000:]}


@pytest.mark.gpu
@pytest.mark.parametrize("name", SECTIONS)
def test_fp16_build(fp16_results, name):
    res = fp16_results["results"]
    assert name in res, f"section did not run (worker exit status {fp16_results['rc']}):\n{fp16_results['tail']}"
    assert res[name] == "ok", res[name]


@pytest.mark.gpu
def test_fp16_worker_ran_every_section(fp16_results):
    assert sorted(fp16_results["results"]) == sorted(SECTIONS), (sorted(fp16_results["results"]), fp16_results["tail"])
    assert fp16_results["rc"] == 0, fp16_results["tail"]


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
