"""Distribution of the "distance travelled" statistic of tests/test_model_gpu.py::test_short_training_tracks_the_fp32_restatement
over seeds: is a bound of 0.5 on |travel ratio - 1| right for the zero-initialised conditional-batch-norm offset tables?

    python scratch/travel_dist.py cpu  OUT.json [n_seeds]      # here (no GPU): the fp32 CPU restatement's side, per seed
    python scratch/travel_dist.py hip  IN.json  OUT.txt        # GPU box: the HIP trainer's side against those summaries

Seed i uses parameter seed 31 + i and data seed 2024 + i (i = 0 is the test's own pair).  The HIP side runs with the
deterministic batch-norm statistics (functional.CONV_EPILOGUE_STATS = False), as the test does.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ITERS, B = 60, 8


def bf16r(a):
    return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)


def feed(rng, it):
    """the test's input stream for one iteration: (g inputs | None, [5 x d inputs])"""
    from oracle import ref_torch as T
    g = None
    if it > 0:
        g = (bf16r(rng.normal(size=(2 * B, 128))), torch.tensor(rng.integers(0, 10, 2 * B), dtype=torch.int32))
    ds = []
    for _ in range(5):
        z = bf16r(rng.normal(size=(B, 128)))
        labels = torch.tensor(rng.integers(0, 10, B), dtype=torch.int32)
        real_u8 = torch.tensor(rng.integers(0, 256, (B, 3072)), dtype=torch.uint8)
        real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(B, 3072, dtype=torch.float64), torch.float64).numpy())
        ds.append((z, labels, real_pre))
    return g, ds


def cpu_side(out, n_seeds):
    from oracle import ref_torch as T
    torch.set_num_threads(int(os.environ.get("THREADS", "4")))
    res = json.load(open(out)) if os.path.exists(out) else {}
    for i in range(n_seeds):
        if str(i) in res:
            continue
        state = T.init_sngan_params(31 + i)
        P = T.to_torch(state, dtype=torch.float32)
        P0 = {k: v.detach().clone() for k, v in P.items()}
        ot = T.Trainer(P)
        rng = np.random.default_rng(2024 + i)
        dl, gl = [], []
        for it in range(ITERS):
            g, ds = feed(rng, it)
            if g is not None:
                gl.append(ot.g_step(it, g[0].to(torch.float32), g[1].long()))
            for z, labels, real_pre in ds:
                dl.append(ot.d_step(it, None, labels.long(), z.to(torch.float32), None, real_pre=real_pre.to(torch.float32)))
        res[str(i)] = {"d_loss": dl, "g_loss": gl,
                       "norm": {k: float(P[k].detach().norm()) for k in ot.g_names + ot.d_names},
                       "norm0": {k: float(P0[k].norm()) for k in ot.g_names + ot.d_names},
                       "numel": {k: int(P[k].numel()) for k in ot.g_names + ot.d_names},
                       "travel": {k: float((P[k].detach() - P0[k]).norm()) for k in ot.g_names + ot.d_names}}
        json.dump(res, open(out, "w"))
        print("cpu seed", i, "done", flush=True)


def hip_side(inp, out):
    from oracle import ref_torch as T
    from gan_lib_tensorflow_amd import functional as Fn
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    Fn.CONV_EPILOGUE_STATS = False
    ref = json.load(open(inp))
    lines = []
    zero_tab, others, norms, dwin, gmean = [], [], [], [], []
    for key in sorted(ref, key=int):
        i, r = int(key), ref[key]
        state = T.init_sngan_params(31 + i)
        tr = S.SNGANTrainer(batch_size=B, seed=31 + i, use_graphs=False, state=state)
        P0 = {k: torch.tensor(v) for k, v in state.items()}
        rng = np.random.default_rng(2024 + i)
        dl, gl = [], []
        for it in range(ITERS):
            g, ds = feed(rng, it)
            if g is not None:
                tr._g_forward_backward(z=g[0].cuda(), fake_labels=g[1].cuda())
                tr._g_apply()
                gl.append(float(tr.g_loss))
            for z, labels, real_pre in ds:
                tr.real_labels.copy_(labels)
                tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
                tr.d_opt.apply()
                dl.append(float(tr.d_loss))
            tr.iteration += 1
            tr.iteration_dev.fill_(tr.iteration)
        dl, gl = np.asarray(dl), np.asarray(gl)
        wd = np.abs(dl.reshape(-1, 50).mean(1) - np.asarray(r["d_loss"]).reshape(-1, 50).mean(1)).max()
        gm = abs(gl.mean() - np.mean(r["g_loss"]))
        tz, to, nr = {}, {}, {}
        for k in r["travel"]:
            if k.startswith('Generator/') and k.endswith('Biases') and 'G.Output' not in k:
                continue
            a = tr.store.vars[k].detach().float().cpu()
            if r["norm"][k] > 1e-3 and r["norm0"][k] > 0.5 * r["norm"][k]:
                nr[k] = abs(float(a.norm()) / r["norm"][k] - 1.0)
            if r["travel"][k] > 1e-3 and r["numel"][k] >= 128:
                v = abs(float((a - P0[k]).norm()) / r["travel"][k] - 1.0)
                (tz if k.endswith('CondBatchNorm/offset') else to)[k] = v
        zero_tab.append(max(tz.values())); others.append(max(to.values())); norms.append(max(nr.values())); dwin.append(wd); gmean.append(gm)
        wk = max(tz, key=tz.get)
        lines.append(f"seed {i:2d}: worst offset-table travel |ratio-1| {max(tz.values()):.3f} ({wk.split('/', 1)[1]}), other tensors {max(to.values()):.3f}, "
                     f"norm ratio {max(nr.values()):.4f}, d-loss window diff {wd:.3f}, g-loss mean diff {gm:.3f}")
        print(lines[-1], flush=True)
        del tr
    q = lambda v: f"min {min(v):.3f}  median {np.median(v):.3f}  p90 {np.quantile(v, 0.9):.3f}  max {max(v):.3f}"   # noqa: E731
    lines += ["", f"{len(zero_tab)} seeds, 60 iterations at batch 8, deterministic batch-norm statistics (CONV_EPILOGUE_STATS = False)",
              f"zero-initialised CondBatchNorm/offset tables, worst travel |ratio-1| per seed: {q(zero_tab)}   (test bound: see tests/test_model_gpu.py)",
              f"every other tensor, worst travel |ratio-1| per seed:                         {q(others)}   (test bound 0.5)",
              f"worst |norm ratio - 1| per seed:                                             {q(norms)}   (test bound 0.02)",
              f"critic-loss 50-update window difference, max per seed:                       {q(dwin)}   (test bound 0.3)",
              f"generator-loss mean difference per seed:                                     {q(gmean)}   (test bound 0.8)"]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-6:]))


if __name__ == "__main__":
    if sys.argv[1] == "cpu":
        cpu_side(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        hip_side(sys.argv[2], sys.argv[3])
