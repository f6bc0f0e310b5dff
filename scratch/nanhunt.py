"""Hunt for intermittent non-finite parameters: run short trainings under several conditions and report the first
iteration / variable that turns non-finite.  POISON=1 fills every torch.empty buffer with NaN first."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

POISON = os.environ.get("POISON", "0") == "1"
if POISON:
    _empty, _empty_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.int32:
            t.fill_(0x7fc07fc0)
        return t

    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t
    torch.empty, torch.empty_like = empty, empty_like

from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S


def first_bad(tr):
    bad = []
    for k, v in tr.store.vars.items():
        if v.is_floating_point() and not bool(torch.isfinite(v).all()):
            bad.append(k)
    return bad


def trial(seed, graphs, iters):
    tr = S.SNGANTrainer(batch_size=64, device="cuda", seed=seed, use_graphs=graphs)
    feed = S.synthetic_batches(64, tr.device, seed=seed)
    for it in range(iters):
        tr.train_iteration(feed)
        torch.cuda.synchronize()
        bad = first_bad(tr)
        gfin = bool(torch.isfinite(tr.g_flat["grads"]).all())
        dfin = bool(torch.isfinite(tr.d_flat["grads"]).all())
        if bad or not gfin or not dfin:
            print(f"  seed {seed} graphs {graphs}: iteration {it}: non-finite vars {bad[:6]} (+{max(0, len(bad) - 6)}) "
                  f"g_grads_finite {gfin} d_grads_finite {dfin} d_loss {float(tr.d_loss):.4f} g_loss {float(tr.g_loss):.4f}", flush=True)
            if not dfin:
                for k, v in tr.store.vars.items():
                    g = getattr(v, "main_grad", None)
                    if g is not None and not bool(torch.isfinite(g).all()):
                        print("     grad non-finite:", k, tuple(v.shape), int((~torch.isfinite(g)).sum()))
            return it
    print(f"  seed {seed} graphs {graphs}: {iters} iterations finite, d_loss {float(tr.d_loss):.4f} g_loss {float(tr.g_loss):.4f}", flush=True)
    return None


if __name__ == "__main__":
    iters = int(os.environ.get("ITERS", "30"))
    print("POISON", POISON, flush=True)
    modes = (True,) if os.environ.get("GRAPHS_ONLY", "0") == "1" else (True, False)
    nbad = 0
    for graphs in modes:
        for seed in range(int(os.environ.get("TRIALS", "4"))):
            nbad += trial(seed, graphs, iters) is not None
    print("FAILED TRIALS:", nbad, flush=True)
