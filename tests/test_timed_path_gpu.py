"""Oracle parity of the path bench.py TIMES, at the headline batch (BASELINE.json config 2, bs = 64), through the captured graphs.

`SNGANTrainer.train_iteration` (SNGAN/gan_cifar_resnet.py:599-620) runs, per iteration: the generator update graph, ONE
320-sample generator pass with 10 statistic groups for the fakes of all five critic updates (`_generate_for_critic`, :326-332 /
:487-495 of the trainer), and five replays of ONE critic graph (`_d_forward_backward_prefetched`) fed by `gank_critic_feed`
from the next slot of the feed ring.  The other network tests drive `_d_forward_backward(real_pre=..., z=...)` eagerly with two
groups; a wrong tower boundary, a stale ring slot, or a replay that reads last update's `u` would pass them.  Here the third
iteration of a graph-mode trainer (every piece a REPLAY) is observed between its pieces (`trainer.observer`) and compared with
`oracle.ref_torch` in float64:

  * the 320 fakes against `generator(..., groups=10)` on the z the device RNG drew (re-drawn from a copy of the state) and the
    iteration's 5 x 64 labels: the headline image bounds (max |d| <= 0.09, mean <= 0.007); the same comparison with the WRONG
    tower split (5 groups of 64) must be clearly worse, i.e. the test sees tower boundaries;
  * each critic update k: its input = [dequantised slot-k reals, slot-k fakes] bit for bit, labels twice; logits and d_loss
    against `discriminator` on that input at the parameters and `u` the update started from (<= 0.02 max(1,|ref|), <= 5e-3); `u`
    afterwards = the oracle's power iteration (the next update starts from it);
  * the gradient the captured optimiser consumed (Adam's `m` slot: beta1 = 0 makes it the gradient itself) of updates 0 and 4
    at the headline per-tensor bounds (cosine >= 0.9999, relative L2 <= 0.015; label-embedding branch 0.999 / 0.03).

And graph replay == eager execution for one whole iteration at bs = 64 from synchronised state, on the deterministic
statistics."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_torch as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gan_lib_tensorflow_amd import kernels
    kernels.lib()
    return torch.device("cuda")


def _grad_errors(got_flat, flat, store, names, ref_g):
    out = {}
    for k in names:
        o, n = flat['offsets'][k], store.vars[k].numel()
        g = got_flat[o:o + n].double().cpu()
        r = ref_g[k].double().flatten()
        rn = float(r.norm())
        out[k] = (float((g @ r) / max(float(g.norm()) * rn, 1e-300)), float((g - r).norm() / max(rn, 1e-300)), rn, float(g.abs().max()))
    return out


def test_timed_path_batch_64_generate_feed_and_five_critic_replays_vs_oracle(gpu):
    from gan_lib_tensorflow_amd import kernels as K
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    seed, b, nc = 77, 64, S.N_CRITIC
    state = T.init_sngan_params(seed)
    tr = S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=True, state=state)
    feed = S.synthetic_batches(b, "cuda", seed=9)
    for _ in range(2):                  # iteration 0 captures 'gen5' and 'd_pre', iteration 1 the generator update
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.use_graphs and {'g', 'gen5', 'd_pre'} <= set(tr._graphs) and tr._graphs['d_pre'][1] is None
    snap = {}

    def observe(stage, k):
        torch.cuda.synchronize()
        if stage == 'before_gen5':
            snap['rng'] = tr.rng_state.clone()
            snap['P_gen'] = tr.store.state_dict()
            snap['labels_all'] = tr.labels_all.clone()
            snap['real_all'] = tr.real_all.clone()
        elif stage == 'after_gen5':
            snap['fake_all'] = tr.fake_all.clone()
            snap['rng_after_gen'] = tr.rng_state.clone()
        elif stage == 'before_d':
            snap['P_d', k] = tr.store.state_dict()
            snap['slot', k] = int(tr.feed_slot)
            snap['t', k] = int(tr.d_opt.t)
        elif stage == 'after_d':
            snap['both', k] = tr.both.clone()
            snap['labels2', k] = tr.both_labels.clone()
            snap['d_loss', k] = float(tr.d_loss)
            snap['logits', k] = tr.last_logits.detach().clone()
            snap['m', k] = tr.d_flat['m'].clone()
    tr.observer = observe
    tr.train_iteration(feed)            # iteration index 2: every piece is a graph replay
    tr.observer = None
    torch.cuda.synchronize()
    assert [snap['slot', k] for k in range(nc)] == list(range(nc)) and int(tr.feed_slot) == 0       # the ring walked 0..4 and wrapped
    assert [snap['t', k] for k in range(nc)] == [10 + k for k in range(nc)]

    # ---- the 320-sample generator pass: 10 towers of 32
    z = K.rng_normal((nc * b, 128), snap['rng'].clone())          # what Generator drew: same counter-based state, same kernel
    assert int(snap['rng_after_gen'][1]) == int(snap['rng'][1]) + 1
    labels_flat = snap['labels_all'].reshape(-1).long().cpu()
    P = T.to_torch(snap['P_gen'])
    with torch.no_grad():
        ref_fake = T.generator(P, z.double().cpu(), labels_flat, groups=nc * S.N_TOWERS)
        wrong = T.generator(P, z.double().cpu(), labels_flat, groups=nc)
    got = snap['fake_all'].reshape(nc * b, 3072).double().cpu()
    diff, diff_wrong = (got - ref_fake).abs(), (got - wrong).abs()
    print("timed path: 320 fakes max/mean |d|", diff.max().item(), diff.mean().item(), " with the wrong tower split:", diff_wrong.mean().item())
    assert diff.max().item() < 0.09 and diff.mean().item() < 0.007, (diff.max().item(), diff.mean().item())
    per_tower = diff.reshape(nc * S.N_TOWERS, -1).mean(1)
    assert per_tower.max().item() < 0.009, per_tower                # no single tower hides behind the average
    assert diff_wrong.mean().item() > 2.0 * diff.mean().item(), (diff_wrong.mean().item(), diff.mean().item())

    # ---- five critic updates, each a replay of the same graph on the next ring slot
    u_names = [k for k in state if k.endswith('spectral_norm/u')]
    dn = None
    for k in range(nc):
        both, labels2 = snap['both', k], snap['labels2', k]
        lab_k = snap['labels_all'][k]
        assert torch.equal(labels2[:b], lab_k) and torch.equal(labels2[b:], lab_k), k
        assert torch.equal(both[b:], snap['fake_all'][k]), k                         # this update's fakes, not another slot's
        base = T.preprocess_real(snap['real_all'][k].cpu(), torch.zeros(b, 3072, dtype=torch.float64), torch.float64)
        noise = both[:b].double().cpu() - base                                      # U[0, 1/128) + bf16 rounding of a value in [-1, 1]
        assert noise.min().item() > -0.0045 and noise.max().item() < 1.0 / 128 + 0.0045, (k, noise.min().item(), noise.max().item())
        assert abs(noise.mean().item() - 1.0 / 256) < 5e-4, (k, noise.mean().item())
        if k > 0:     # the dequantisation noise is drawn anew per update (the RNG offset advanced inside the previous replay)
            assert not torch.equal(both[:b], snap['both', k - 1][:b])
        Pk = T.to_torch(snap['P_d', k])
        ref_logits, new_u = T.discriminator(Pk, both.double().cpu(), labels2.long().cpu())
        loss = torch.relu(1. - ref_logits[:b]).mean() + torch.relu(1. + ref_logits[b:]).mean()
        dl = (snap['logits', k].double().cpu().reshape(-1) - ref_logits.detach()).abs()
        print(f"timed path: update {k}: logits max |d| {dl.max().item():.2e}  d_loss {snap['d_loss', k]:.5f} vs {float(loss):.5f}")
        assert (dl <= 0.02 * torch.clamp(ref_logits.detach().abs(), min=1.0)).all(), (k, dl.max().item())
        assert abs(snap['d_loss', k] - float(loss)) < 5e-3, (k, snap['d_loss', k], float(loss))
        assert float(loss) > 0.2, "hinge saturated: the gradient comparison below would be vacuous"
        after = snap['P_d', k + 1] if k + 1 < nc else tr.store.state_dict()
        for un in u_names:               # update_collection=None: u <- the power iteration's result (sn.py:48-56)
            r = new_u[un].reshape(-1)
            d = np.abs(after[un].reshape(-1) - r.numpy()).max() / max(float(r.abs().max()), 1e-30)
            assert d < 1e-3, (k, un, d)
        if k in (0, nc - 1):
            dn = T.trainable_names(Pk, 'Discriminator')
            ref_g = dict(zip(dn, torch.autograd.grad(loss, [Pk[n] for n in dn])))
            errs = _grad_errors(snap['m', k], tr.d_flat, tr.store, dn, ref_g)
            print(f"timed path: update {k} gradients (cos, relL2):", {n.split('/', 1)[1]: (round(c, 5), round(l, 4)) for n, (c, l, _, _) in errs.items()})
            bad = []
            for n, (cos, l2, rn, gmax) in errs.items():
                if rn < 1e-9:
                    if gmax > 1e-6:
                        bad.append((n, 'abs', gmax))
                    continue
                lim = (0.999, 0.03) if 'mbedding' in n else (0.9999, 0.015)
                if cos < lim[0] or l2 > lim[1]:
                    bad.append((n, cos, l2))
            assert not bad, (k, bad)


def _sync_trainers(dst, src):
    """every mutable piece of trainer state, written IN PLACE (captured graphs keep their addresses)"""
    with torch.no_grad():
        for k, v in src.store.vars.items():
            dst.store.vars[k].copy_(v)
        for net in ('Generator', 'Discriminator'):
            for key in ('m', 'v'):
                dst.store.flat[net][key].copy_(src.store.flat[net][key])
        for a, c in ((dst.g_opt, src.g_opt), (dst.d_opt, src.d_opt)):
            a.t.copy_(c.t)
        dst.rng_state.copy_(src.rng_state)
        dst.iteration_dev.copy_(src.iteration_dev)
        dst.feed_slot.copy_(src.feed_slot)
        dst.iteration = src.iteration
        for tr in (dst, src):
            for flat in (tr.g_flat, tr.d_flat):
                flat['grads_all'].zero_()
                flat['clean'] = True
    dst._refresh_g_prep()
    dst.sn_state_changed()           # the critic's weights and u were written behind the trainer's back
    torch.cuda.synchronize()


def test_graph_replay_equals_eager_execution_for_one_iteration_at_batch_64(gpu, deterministic_stats):
    """One whole `train_iteration` at bs = 64 from bit-identical state: an eager trainer against a graph-mode trainer whose every
    piece is a replay, on the deterministic statistics.  The only run-to-run freedom left is the arrival order of the fp32
    atomics in the filter-gradient / table-gradient sums, so:
      * the generator update: identical RNG consumption, same loss, parameters equal except for sign flips of ~0 gradients
        (TF-Adam with beta1 = 0 turns those into 2 lr);
      * the 320-sample generator pass has NO atomics: from bit-identical weights (the replay trainer takes over the eager
        trainer's generator parameters before it) the 320 fakes are BIT-IDENTICAL;
      * the first critic update then sees bit-identical inputs and parameters: logits and loss bit-identical; updates 1..4
        inherit the atomics-order noise of the updates before them: bounds grow with k."""
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    seed, b, nc = 78, 64, S.N_CRITIC
    state = T.init_sngan_params(seed)
    tr_e = S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=False, state=state)
    tr_g = S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=True, state=state)
    feed_e, feed_g = S.synthetic_batches(b, "cuda", seed=3), S.synthetic_batches(b, "cuda", seed=3)
    for _ in range(2):
        tr_e.train_iteration(feed_e)
        tr_g.train_iteration(feed_g)
    torch.cuda.synchronize()
    assert tr_g.use_graphs and {'g', 'gen5', 'd_pre'} <= set(tr_g._graphs)
    _sync_trainers(tr_g, tr_e)
    rec = {id(tr_e): {}, id(tr_g): {}}

    def observer(tr):
        def fn(stage, k):
            torch.cuda.synchronize()
            r = rec[id(tr)]
            if stage == 'after_g':
                r['g_params'] = tr.g_flat['params'].clone()
                r['g_loss'] = float(tr.g_loss)
            elif stage == 'before_gen5' and tr is tr_g:
                # the eager trainer has finished its whole iteration by now: take over ITS generator after the update
                with torch.no_grad():
                    tr.g_flat['params'].copy_(rec[id(tr_e)]['g_params'])
                tr._refresh_g_prep()
                torch.cuda.synchronize()
            elif stage == 'after_gen5':
                r['fake_all'] = tr.fake_all.clone()
            elif stage == 'after_d':
                r['d_loss', k] = float(tr.d_loss)
                r['logits', k] = tr.last_logits.detach().float().clone()
                r['d_params', k] = tr.d_flat['params'].clone()
        return fn
    tr_e.observer, tr_g.observer = observer(tr_e), observer(tr_g)
    tr_e.train_iteration(feed_e)
    tr_g.train_iteration(feed_g)
    torch.cuda.synchronize()
    tr_e.observer = tr_g.observer = None
    e, g = rec[id(tr_e)], rec[id(tr_g)]
    assert torch.equal(tr_e.rng_state, tr_g.rng_state) and int(tr_e.feed_slot) == int(tr_g.feed_slot) == 0
    assert int(tr_e.d_opt.t) == int(tr_g.d_opt.t) == 15 and int(tr_e.g_opt.t) == int(tr_g.g_opt.t) == 2
    assert abs(e['g_loss'] - g['g_loss']) < 1e-3, (e['g_loss'], g['g_loss'])
    d = (e['g_params'] - g['g_params']).abs()
    print("eager vs replay: G params after one update: frac > 2e-5:", (d > 2e-5).float().mean().item(), "mean", d.mean().item())
    assert (d > 2e-5).float().mean().item() < 5e-3 and d.mean().item() < 2e-6, ((d > 2e-5).float().mean().item(), d.mean().item())     # measured 2.9e-4 / 2.0e-7
    df = (e['fake_all'].float() - g['fake_all'].float()).abs()
    print("eager vs replay: 320 fakes from identical weights: max |d|", df.max().item(), "differing elements", int((df > 0).sum()))
    assert torch.equal(e['fake_all'], g['fake_all']), (df.max().item(), int((df > 0).sum()))
    for k in range(nc):
        dl = (e['logits', k] - g['logits', k]).abs().max().item()
        dp = (e['d_params', k] - g['d_params', k]).abs()
        print(f"eager vs replay: update {k}: d_loss {e['d_loss', k]:.6f} / {g['d_loss', k]:.6f}  logits max |d| {dl:.3e}  "
              f"params frac > 2e-5 {(dp > 2e-5).float().mean().item():.2e} mean {dp.mean().item():.2e}")
        if k == 0:
            assert dl == 0.0 and e['d_loss', 0] == g['d_loss', 0], (dl, e['d_loss', 0], g['d_loss', 0])
        assert abs(e['d_loss', k] - g['d_loss', k]) < 5e-3 * k + 1e-6, (k, e['d_loss', k], g['d_loss', k])
        assert dl < 0.05 * k + 1e-6, (k, dl)
        assert dp.mean().item() < 1e-5 * (k + 1) and dp.max().item() < (k + 2) * 2 * 2e-4, (k, dp.mean().item(), dp.max().item())
