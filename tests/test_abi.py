"""CPU: the C-ABI library builds, loads, and exports every symbol include/gank.h declares (and the
ctypes table binds exactly that set).  No compute calls -- there is no GPU here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "gank.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gank_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    from gan_lib_tensorflow_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in gank.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, set(_lib.PROTOTYPES) ^ set(syms)
    assert lib.gank_version() == 100


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan_lib_tensorflow_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(d, f)


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gan_lib_tensorflow_amd import kernels as K
    x = torch.zeros(8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        K.relu_fwd(x)


def test_entry_points_reject_bad_arguments_with_a_message():
    """Error behaviour of the C ABI: a non-zero return and a gank_last_error() message, decided on the host before
    anything is launched (so this runs without a GPU): null pointers, empty tables, unsupported shapes."""
    import ctypes as C
    from gan_lib_tensorflow_amd import _lib
    lib = _lib.load()
    lib.gank_last_error.restype = C.c_char_p
    P = C.c_void_p
    fake = P(0x1000)      # never dereferenced: every call below is refused by argument checks

    def err():
        return lib.gank_last_error().decode()

    assert lib.gank_conv2d_fprop(None, None, None, None, None, None, 1, 8, 8, 64, 64, 3, 0, C.c_float(1.0), None) != 0
    assert "null pointer" in err()
    assert lib.gank_conv2d_fprop(fake, fake, None, None, None, fake, 1, 8, 8, 64, 64, 2, 0, C.c_float(1.0), None) != 0
    assert "even filter sizes" in err()
    assert lib.gank_conv2d_fprop(fake, fake, None, None, None, fake, 1, 7, 7, 64, 64, 3, 1, C.c_float(1.0), None) != 0   # IN_UPSAMPLE2X, odd size
    assert "even output size" in err()
    assert lib.gank_conv2d_fprop(fake, fake, None, None, None, fake, 1, 8, 8, 64, 64, 3, 64, C.c_float(1.0), None) != 0  # RES_UPSAMPLE2X without residual
    assert "needs a residual" in err()
    assert lib.gank_conv2d_wgrad(fake, fake, fake, None, None, 0, 1, 8, 8, 64, 64, 4, 0, C.c_float(1.0), None) != 0
    assert "even filter sizes" in err()
    assert lib.gank_sn_power_iter_fwd(None, 0, None) != 0 and "empty table" in err()
    assert lib.gank_conv2d_prep_weights_batched(None, 0, None) != 0 and "empty table" in err()
    desc = (_lib.PrepDesc * 1)()
    desc[0].w, desc[0].wf, desc[0].wd = 0x1000, 0x1000, 0x1000
    desc[0].ksize, desc[0].Cin, desc[0].Cout, desc[0].kind = 1, 64, 64, 2        # kind 2 needs ksize 3
    assert lib.gank_conv2d_prep_weights_batched(desc, 1, None) != 0 and "kind 2" in err()
    assert lib.gank_cbn_fwd(fake, fake, fake, fake, fake, fake, fake, 4, 16, 12, 1, 10, 0, None) != 0          # C = 12
    assert "unsupported" in err()
    assert lib.gank_convpool3x3_dgrad(fake, fake, None, fake, 1, 8, 8, 128, 32, None) != 0
    assert "multiple of 64" in err()
    assert lib.gank_linear_bwd(fake, None, None, fake, None, None, 4, 8, 8, None) != 0 and "dx needs w" in err()
    assert lib.gank_copy_bytes(fake, fake, 0, None) != 0 and "bad arguments" in err()
    assert lib.gank_copy_bytes_gather(fake, fake, 17, 16, None) != 0 and "1..16 sources" in err()
    assert lib.gank_conv2d_wgrad_batched(None, 0, 1, 8, 8, 128, 128, 3, 0, C.c_float(1.0), None) != 0 and "empty list" in err()
