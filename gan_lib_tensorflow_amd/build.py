"""Build libgank.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m gan_lib_tensorflow_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, os.environ.get("GANK_LIB_NAME", "libgank.so"))   # experiment builds: GANK_LIB_NAME + GANK_EXTRA_FLAGS
SOURCES = ["api.hip", "conv_igemm.hip", "conv_wgrad.hip", "conv_resident.hip", "label_conv.hip", "acgan_ops.hip", "pggan_pix_ops.hip", "elementwise.hip", "sn.hip", "cbn.hip", "loss_opt.hip", "linear.hip", "norms.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("GANK_EXTRA_FLAGS", "").split()


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, variants=("bf16", "fp16")):
    """libgank.so (bfloat16 buffers) and libgank_f16.so (IEEE half: the same sources with -DGANK_ACT_F16); returns the first
    path.  Every out-of-date object of BOTH variants compiles concurrently (one hipcc process per source file)."""
    base_flags = [f for f in FLAGS if f != "-DGANK_ACT_F16"]
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INC, "gank.h")]   # every header rebuilds every object
    plans = []
    for variant in variants:
        if variant == "fp16":
            if "GANK_LIB_NAME" in os.environ or os.environ.get("GANK_EXTRA_FLAGS"):
                continue        # an experiment build (scratch/README.md) is the bf16 library under its own name: the production
                                # fp16 objects must not be rebuilt with the experiment's flags
            lib, flags, objdir = os.path.join(HERE, "libgank_f16.so"), base_flags + ["-DGANK_ACT_F16"], "_obj_f16"
        else:
            lib, flags = os.path.join(HERE, os.environ.get("GANK_LIB_NAME", "libgank.so")), base_flags
            objdir = "_obj" + ("_" + os.environ["GANK_LIB_NAME"] if "GANK_LIB_NAME" in os.environ else "")
        os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
        # a change of flags rebuilds every object of the directory (mtimes alone would keep objects of the old flags)
        stamp, flag_str = os.path.join(HERE, objdir, "flags.txt"), " ".join(flags)
        stale = not os.path.exists(stamp) or open(stamp).read() != flag_str
        if stale and os.path.exists(stamp):
            os.remove(stamp)        # rewritten only after every compile of this directory has succeeded: an interrupted
                                    # or failed build must not leave old-flag objects that the next build takes as current
        objs, procs = [], []
        for src in SOURCES:
            sp = os.path.join(CSRC, src)
            op = os.path.join(HERE, objdir, src.replace(".hip", ".o"))
            objs.append(op)
            if force or stale or _newer(sp, op) or any(_newer(h, op) for h in hdrs):
                cmd = ["hipcc", *flags, "-I", INC, "-c", sp, "-o", op]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        plans.append((lib, objs, procs, stamp, flag_str))
    failed = False
    for _, _, procs, stamp, flag_str in plans:
        plan_failed = False
        for src, p in procs:
            out = p.communicate()[0].decode()
            if out.strip() and verbose:
                print(out)
            if p.returncode != 0:
                print(out, file=sys.stderr)
                failed = plan_failed = True
        if not plan_failed:
            with open(stamp, "w") as f:
                f.write(flag_str)
    if failed:
        raise RuntimeError("hipcc failed")
    for lib, objs, procs, _, _ in plans:
        if force or procs or not os.path.exists(lib) or any(_newer(o, lib) for o in objs):     # (a link that failed after its objects compiled)
            cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return plans[0][0]


if __name__ == "__main__":
    build(force="--force" in sys.argv)
