"""Network-level parity on a real MI355X: Generator / Discriminator / train step (HIP path through
the C ABI) vs the torch-CPU float64 oracle on identical parameters and inputs, plus the committed
golden vectors.  Tolerances are for the bf16 compute path against a float64 restatement of the
fp32 reference: images (tanh range) max |d| <= 0.08 and mean |d| <= 0.008 (measured: 0.045 / 0.005 -- four
conditional-batch-norm stages over two samples per tower amplify bf16 storage rounding); logits
|d| <= 0.06*max(1,|ref|) (measured 1e-3); gradients per tensor by cosine >= 0.97..0.98 and relative L2 (stated at
each assertion); eager vs hipGraph: tight for the first two updates, sanity-only afterwards (fp32 atomics make
the trajectories diverge chaotically at the bf16 rounding level)."""
import gc
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


def bf16r(a):
    return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)


def make_trainer(seed, batch, use_graphs=False):
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    state = T.init_sngan_params(seed)
    tr = S.SNGANTrainer(batch_size=batch, seed=seed, use_graphs=use_graphs, state=state)
    return S, tr, state


def rel(got, ref):
    got = got.detach().to(torch.float64).cpu().numpy()
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all()
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def test_names_and_param_counts(gpu):
    S, tr, state = make_trainer(0, 4)
    assert sorted(tr.store.vars.keys()) == sorted(state.keys())
    assert set(tr.store.vars.keys()) == set(state.keys())
    assert tr.store.param_count('Generator') == 7875587
    assert tr.store.param_count('Discriminator') == 1701689
    assert len([k for k in tr.store.vars if k.endswith('spectral_norm/u')]) == 12


def test_generator_discriminator_forward_and_golden(gpu, golden_dir):
    gold = np.load(os.path.join(golden_dir, "network.npz"))
    S, tr, state = make_trainer(int(gold["seed"]), 4)
    z = bf16r(gold["z"])
    labels = torch.tensor(gold["labels"], dtype=torch.int32)
    with torch.no_grad():
        img = S.Generator(4, labels.cuda(), noise=z.cuda(), groups=2)
    P = T.to_torch(state)
    ref_img = T.generator(P, z.to(torch.float64), labels.long(), groups=2).detach()
    diff = (img.to(torch.float64).cpu() - ref_img).abs()
    # 4 CBN stages with statistics over only 2 samples x 16..1024 pixels amplify bf16 rounding: bound the
    # worst pixel loosely and the mean tightly
    assert diff.max().item() < 0.08 and diff.mean().item() < 0.008, (diff.max().item(), diff.mean().item())
    assert np.abs(img.to(torch.float64).cpu().numpy()[:, :96] - gold["img_head"]).max() < 0.08
    # critic on the oracle's (bf16-rounded) inputs, update_collection=None
    real = T.preprocess_real(torch.tensor(gold["real_u8"]), torch.zeros(4, 3072, dtype=torch.float64), torch.float64)
    both = torch.cat([bf16r(real.numpy()), bf16r(ref_img.numpy())], 0)
    both_labels = torch.cat([labels, labels])
    u_before = tr.store.vars['Discriminator/D.Output/spectral_norm/u'].clone()
    with torch.no_grad():
        logits, _ = S.Discriminator(both.cuda(), both_labels.cuda(), update_collection=S.NO_OPS)
    assert torch.equal(u_before, tr.store.vars['Discriminator/D.Output/spectral_norm/u'])       # NO_OPS: never written
    ref_logits = gold["logits"]
    d = np.abs(logits.to(torch.float64).cpu().numpy() - ref_logits)
    assert (d <= 0.06 * np.maximum(1., np.abs(ref_logits))).all(), (d, ref_logits)
    with torch.no_grad():
        S.Discriminator(both.cuda(), both_labels.cuda(), update_collection=None)
    u_after = tr.store.vars['Discriminator/D.Output/spectral_norm/u']
    assert rel(u_after, gold["u_new_D_Output"]) < 1e-4                                             # None: overwritten


def test_d_and_g_gradients_vs_oracle(gpu):
    seed, b = 5, 4
    S, tr, state = make_trainer(seed, b)
    rng = np.random.default_rng(7)
    z = bf16r(rng.normal(size=(b, 128)))
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy())
    P = T.to_torch(state)
    # ---- D loss gradients
    loss, _, ref_logits = T.d_loss_fn(P, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    dn = T.trainable_names(P, 'Discriminator')
    ref_g = dict(zip(dn, torch.autograd.grad(loss, [P[k] for k in dn])))
    tr.real_labels.copy_(labels)
    logits = tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.d_loss) - float(loss)) < 0.05
    errs = {k: rel(tr.store.vars[k].main_grad, ref_g[k].numpy()) for k in dn}
    print("D grad rel errors:", {k.split('/', 1)[1]: round(e, 4) for k, e in errs.items()})
    # the label-embedding branch is a sum over 16x16 pixels of bf16 activation gradients that largely
    # cancel (the embedding is constant over pixels): its relative error is amplified -> looser bound
    bad = [(k, e) for k, e in errs.items() if e > (0.2 if 'mbedding' in k else 0.1)]
    assert not bad, bad
    # max-norm error is dominated by a few relu-mask flips of near-zero bf16 activations; the L2 error and the
    # direction are the robust statement (measured: L2 1-4 %, cosine > 0.999 on the conv filters)
    for k in dn:
        g = tr.store.vars[k].main_grad.double().cpu().flatten()
        r = ref_g[k].flatten()
        if r.norm() < 1e-12:
            continue
        cos = float((g @ r) / (g.norm() * r.norm()))
        l2 = float((g - r).norm() / r.norm())
        lim = (0.98, 0.2) if 'mbedding' in k else (0.995, 0.08)
        assert cos > lim[0] and l2 < lim[1], (k, cos, l2)
    # ---- G loss gradients (fresh oracle params: the critic's u was advanced by the D pass above)
    P = T.to_torch(tr.store.state_dict())
    z2 = bf16r(rng.normal(size=(2 * b, 128)))
    fl = torch.tensor(rng.integers(0, 10, 2 * b), dtype=torch.int32)
    loss, _ = T.g_loss_fn(P, z2.to(torch.float64), fl.long())
    gn = T.trainable_names(P, 'Generator')
    ref_g = dict(zip(gn, torch.autograd.grad(loss, [P[k] for k in gn])))
    tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.g_loss) - float(loss)) < 0.05
    # Generator gradients travel back through 7 conditional batch norms whose backward subtracts batch
    # means from bf16-stored gradients (cancellation amplifies the 2^-9 storage rounding; statistics here
    # are over only 4 samples per tower).  Bound direction and L2 error per tensor.  Conv biases that
    # feed a batch norm have an exactly-zero true gradient: bound them absolutely.
    bad = []
    for k in gn:
        g = tr.store.vars[k].main_grad.double().cpu().flatten()
        r = ref_g[k].flatten()
        if k.endswith('Biases') and 'G.Output' not in k:
            if g.abs().max() > 1e-3:
                bad.append((k, 'abs', float(g.abs().max())))
            continue
        cos = float((g @ r) / (g.norm() * r.norm()))
        l2 = float((g - r).norm() / r.norm())
        if cos < 0.98 or l2 > 0.2:
            bad.append((k, cos, l2))
    assert not bad, bad


@pytest.mark.parametrize("loss_type,soft_plus", [("Goodfellow", False), ("WGAN", False), ("Goodfellow", True), ("HINGE", True), ("WGAN", True)])
def test_other_loss_types_of_the_script_vs_oracle(gpu, loss_type, soft_plus):
    """LOSS_TYPE of SNGAN/gan_cifar_resnet.py:62: 'Goodfellow' (:363-369 critic, :483-486 generator) and 'WGAN' (:382-387, :493-497),
    and the SOFT_PLUS = True (:63) variants of all three types, through `SNGANTrainer(loss_type=..., soft_plus=...)`: both losses and
    every gradient against torch-float64 autograd of the script's expressions (oracle.ref_torch.sngan_losses) on the oracle's
    logits (tolerances of test_d_and_g_gradients_vs_oracle); an unknown type raises as a missing branch would."""
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    seed, b = 6, 4
    state = T.init_sngan_params(seed)
    tr = S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=False, state=state, loss_type=loss_type, soft_plus=soft_plus)
    rng = np.random.default_rng(17)
    z = bf16r(rng.normal(size=(b, 128)))
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy())
    P = T.to_torch(state)
    _, _, lg = T.d_loss_fn(P, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    loss, _ = T.sngan_losses(lg, b, None, loss_type, soft_plus)
    dn = T.trainable_names(P, 'Discriminator')
    ref_g = dict(zip(dn, torch.autograd.grad(loss, [P[k] for k in dn])))
    tr.real_labels.copy_(labels)
    tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.d_loss) - float(loss)) < 0.05, (float(tr.d_loss), float(loss))
    for k in dn:
        g = tr.store.vars[k].main_grad.double().cpu().flatten()
        r = ref_g[k].flatten()
        if r.norm() < 1e-12:
            continue
        cos, l2 = float((g @ r) / (g.norm() * r.norm())), float((g - r).norm() / r.norm())
        lim = (0.98, 0.2) if 'mbedding' in k else (0.995, 0.08)
        assert cos > lim[0] and l2 < lim[1], (k, cos, l2)
    P = T.to_torch(tr.store.state_dict())
    z2 = bf16r(rng.normal(size=(2 * b, 128)))
    fl = torch.tensor(rng.integers(0, 10, 2 * b), dtype=torch.int32)
    _, lg = T.g_loss_fn(P, z2.to(torch.float64), fl.long())
    _, loss = T.sngan_losses(None, 0, lg, loss_type, soft_plus)
    gn = T.trainable_names(P, 'Generator')
    ref_g = dict(zip(gn, torch.autograd.grad(loss, [P[k] for k in gn])))
    tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.g_loss) - float(loss)) < 0.05
    bad = []
    for k in gn:
        g, r = tr.store.vars[k].main_grad.double().cpu().flatten(), ref_g[k].flatten()
        if k.endswith('Biases') and 'G.Output' not in k:
            continue
        cos, l2 = float((g @ r) / (g.norm() * r.norm())), float((g - r).norm() / r.norm())
        if cos < 0.98 or l2 > 0.2:
            bad.append((k, cos, l2))
    assert not bad, bad
    with pytest.raises(NotImplementedError):
        S.SNGANTrainer(batch_size=b, seed=seed, use_graphs=False, loss_type='WGAN-GP')


def test_train_steps_eager_vs_graph_and_oracle_update(gpu):
    """Same seeds, same feed: hipGraph replay must reproduce the eager steps (bit-for-bit except the
    fp32 atomics of wgrad), and one D update must move the parameters like the oracle's TF-Adam."""
    from gan_lib_tensorflow_amd import functional as Fn
    # the batch-norm statistics of this part come from the deterministic three-pass kernels: sums accumulated by conv
    # epilogues (fp32 atomics, arrival order) make the FAKES differ in their last bf16 bit from run to run, and this
    # comparison isolates what graph replay itself changes
    stats_were, Fn.CONV_EPILOGUE_STATS = Fn.CONV_EPILOGUE_STATS, False
    S, tr_e, state = make_trainer(11, 8, use_graphs=False)
    feed_e = S.synthetic_batches(8, "cuda", seed=1)
    _, tr_g, _ = make_trainer(11, 8, use_graphs=True)
    feed_g = S.synthetic_batches(8, "cuda", seed=1)
    # first two critic updates: step 1 runs eagerly in both, step 2 is a graph REPLAY in tr_g.  Only the
    # order of the fp32 wgrad atomics differs, so parameters agree to ~1e-7 (measured 2.4e-7).
    for _ in range(2):
        tr_e.d_step(*next(feed_e))
        tr_g.d_step(*next(feed_g))
    torch.cuda.synchronize()
    assert torch.equal(tr_e.rng_state, tr_g.rng_state)          # replay consumed the device RNG like eager
    a, b = tr_e.d_flat["params"], tr_g.d_flat["params"]
    # (TF-Adam normalises by sqrt(v): a ~0 gradient that differs by 1e-8 from atomic ordering moves its weight
    # by a different fraction of lr=2e-4, or flips its sign in the first step -- allow a few such entries)
    d = (a - b).abs()
    assert (d > 2e-5).float().mean().item() < 5e-3 and d.mean().item() < 2e-6, ((d > 2e-5).float().mean().item(), d.mean().item())
    assert abs(float(tr_e.d_loss) - float(tr_g.d_loss)) < 1e-4
    Fn.CONV_EPILOGUE_STATS = stats_were
    tr_e._graphs.clear()
    tr_g._graphs.clear()
    # then full iterations.  TF-Adam with beta1=0 moves a weight by ~lr*sign(g) on its first step, so an
    # atomics-order flip of a ~0 gradient is a 2*lr jump and bf16 rounding boundaries amplify it from
    # there (two EAGER runs diverge the same way): later steps are compared statistically.
    for _ in range(3):
        tr_e.train_iteration(feed_e)
        tr_g.train_iteration(feed_g)
    torch.cuda.synchronize()
    assert tr_e.iteration == 3 and int(tr_e.iteration_dev) == 3 and int(tr_e.d_opt.t) == 17 and int(tr_e.g_opt.t) == 2
    assert int(tr_g.iteration_dev) == 3 and int(tr_g.d_opt.t) == 17 and int(tr_g.g_opt.t) == 2
    for net in ('Generator', 'Discriminator'):
        a, b = tr_e.store.flat[net]["params"], tr_g.store.flat[net]["params"]
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        # the two trajectories have diverged chaotically by now (see above): only sanity is asserted, each
        # parameter can have moved by at most ~lr per update
        assert (a - b).abs().max().item() < 60 * 2e-4
    # hinge loss of a young critic on two diverged trajectories: only its range is asserted
    assert 0.0 <= float(tr_e.d_loss) < 4.0 and 0.0 <= float(tr_g.d_loss) < 4.0

    # one D update vs the oracle's TF-Adam from the same state
    S, tr, state = make_trainer(13, 4, use_graphs=False)
    rng = np.random.default_rng(3)
    z = bf16r(rng.normal(size=(4, 128)))
    labels = torch.tensor(rng.integers(0, 10, 4), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (4, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(4, 3072, dtype=torch.float64), torch.float64).numpy())
    P = T.to_torch(state)
    ot = T.Trainer(P)
    ot.d_step(0, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    tr.real_labels.copy_(labels)
    tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
    tr.d_opt.apply()
    torch.cuda.synchronize()
    for k in ('Discriminator/D.Block.3.Conv1/Filters', 'Discriminator/D.Output/W', 'Discriminator/D.Block.1.Conv1/Biases'):
        got = tr.store.vars[k].detach().to(torch.float64).cpu()
        d = (got - P[k].detach()).abs()
        # first TF-Adam step with beta1=0 moves every weight by ~lr*sign(g): allow sign flips of ~0 gradients
        assert (d > 0.5 * 2e-4).float().mean().item() < 0.05, k
    assert rel(tr.store.vars['Discriminator/D.Output/spectral_norm/u'], P['Discriminator/D.Output/spectral_norm/u'].numpy()) < 1e-3


def test_sampling_path(gpu):
    S, tr, _ = make_trainer(2, 4)
    img = tr.sample(100)
    assert img.shape == (100, 3072) and float(img.abs().max()) <= 1.0


def test_fixed_noise_samples_batch_100_vs_oracle(gpu):
    """The sample grid of gan_cifar_resnet.py:530-538: Generator(100, [0..9]*10, noise=fixed_noise) -- a batch that is
    neither a power of two nor a multiple of 64 through every kernel, conditional-batch-norm statistics over the 100
    samples -- against the float64 oracle, and its host-side quantisation."""
    from gan_lib_tensorflow_amd.common.inception.inception_score import quantize_samples
    S, tr, state = make_trainer(11, 4)
    rng = np.random.default_rng(100)
    z = bf16r(rng.normal(size=(100, 128)))
    labels = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, 9] * 10, dtype=torch.int32)
    img = tr.sample(100, labels=labels.cuda(), noise=z.cuda())
    ref = T.generator(T.to_torch(state), z.to(torch.float64), labels.long(), groups=1).detach()
    diff = (img.to(torch.float64).cpu() - ref).abs()
    # same bounds as the 4-sample golden case above: bf16 storage between 20 layers, fp32 statistics
    assert diff.max().item() < 0.08 and diff.mean().item() < 0.008, (diff.max().item(), diff.mean().item())
    q, qr = quantize_samples(img.float().cpu().numpy()), quantize_samples(ref.numpy())
    assert q.dtype == np.int32 and q.min() >= 0 and q.max() <= 255
    assert np.abs(q - qr).mean() < 1.0                      # under one grey level on average


def test_graph_replay_stays_finite_with_poisoned_workspaces(gpu, monkeypatch):
    """Every captured update must fully define what it reads: with every torch.empty buffer pre-filled with NaN,
    several trainers replaying their hipGraphs keep finite parameters (regression: a hipMemsetAsync memset NODE
    in the conditional-batch-norm backward lost its ordering under replay and let NaN workspaces through)."""
    _empty, _empty_like = torch.empty, torch.empty_like

    def empty(*a, **k):
        t = _empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    monkeypatch.setattr(torch, "empty", empty)
    monkeypatch.setattr(torch, "empty_like", empty_like)
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    for seed in range(3):
        tr = S.SNGANTrainer(batch_size=64, device="cuda", seed=seed, use_graphs=True)
        feed = S.synthetic_batches(64, tr.device, seed=seed)
        for _ in range(6):
            tr.train_iteration(feed)
        torch.cuda.synchronize()
        assert tr.use_graphs, "graph capture fell back to eager"
        for flat in (tr.g_flat, tr.d_flat):
            assert bool(torch.isfinite(flat["params"]).all()) and bool(torch.isfinite(flat["grads"]).all())
        del tr, feed
        gc.collect()
        torch.cuda.synchronize()


def test_unconditional_generator_uses_batch_norm(gpu):
    """CONDITIONAL=False routes Normalize to batch_norm (gan_cifar_resnet.py:92-106): with freshly initialised
    tables (scale 1, offset 0) the conditional and the unconditional generator are the same function, and the
    unconditional one owns BatchNorm variables with moving statistics instead of CondBatchNorm tables."""
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    from gan_lib_tensorflow_amd.common import resnet_block as blocks
    from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
    g = torch.Generator().manual_seed(4)
    z = torch.randn(8, 128, generator=g).to(torch.bfloat16).cuda()
    labels = torch.randint(0, 10, (8,), generator=g, dtype=torch.int32).cuda()
    imgs = {}
    from gan_lib_tensorflow_amd import functional as Fn
    stats_were = Fn.CONV_EPILOGUE_STATS
    for cond in (True, False):
        blocks.CONDITIONAL = cond
        # both paths on the three-pass statistics kernels: the bit-for-bit comparison below is about the routing, and
        # statistics accumulated by conv epilogues (conditional path only) round differently in their last bit
        Fn.CONV_EPILOGUE_STATS = False
        try:
            store = set_default_store(ParamStore("cuda", seed=7))
            z_in = z.clone().requires_grad_(not cond)
            img = S.Generator(8, labels, noise=z_in, groups=2)
            imgs[cond] = img.detach().clone()
            names = list(store.vars)
            if cond:
                assert any(k.endswith('CondBatchNorm/scale') for k in names) and not any('BatchNorm/gamma' in k for k in names)
            else:
                assert any(k.endswith('G.Block.1.N1/BatchNorm/gamma') for k in names) and not any('CondBatchNorm' in k for k in names)
                img.float().sum().backward()                      # gradients flow through the batch-norm path
                gm = store.vars['Generator/G.OutputNorm/BatchNorm/gamma']
                assert gm.grad is not None and bool(torch.isfinite(gm.grad).all()) and float(gm.grad.abs().sum()) > 0
                assert float(store.vars['Generator/G.OutputNorm/BatchNorm/moving_mean/local_step']) == 2.0    # two towers
        finally:
            blocks.CONDITIONAL = True
            Fn.CONV_EPILOGUE_STATS = stats_were
    assert torch.equal(imgs[True], imgs[False])


def _grad_errors(tr, names, ref_g):
    out = {}
    for k in names:
        g = tr.store.vars[k].main_grad.double().cpu().flatten()
        r = ref_g[k].double().flatten()
        rn = float(r.norm())
        out[k] = (float((g @ r) / max(float(g.norm()) * rn, 1e-300)), float((g - r).norm() / max(rn, 1e-300)), rn, float(g.abs().max()))
    return out


def test_headline_batch_64_forward_and_gradients_vs_oracle(gpu):
    """BASELINE.json config 2 at its real size -- batch 64 (32 samples per tower), critic on 64 real + 64 fake, generator
    update on 2 x 64 fakes -- against the float64 oracle.  At this size the conditional-batch-norm statistics are
    over 32 x 16..1024 values per channel.  Stated bounds (measured values in brackets):
      * critic: logits <= 0.02*max(1,|ref|) [8e-4], loss <= 5e-3 [2.4e-4], every gradient tensor cosine >= 0.9999 and
        relative L2 <= 0.015 [0.003..0.007]; the label-embedding branch (gradients constant over 16 x 16 pixels that
        largely cancel) cosine >= 0.999, L2 <= 0.03 [0.011];
      * generator images (tanh range): max |d| <= 0.09, mean <= 0.007 [0.060 / 0.0045]: 20 bf16-stored layers, the same
        as at the toy batch -- the error is storage rounding, not statistics;
      * generator gradients: relative L2 grows by 1-2 % per conditional batch norm on the way back from the image
        [G.Output 0.003, G.Block.3 0.03-0.06, G.Block.2 0.04-0.08, G.Block.1 / G.Input 0.07-0.145], cosine >= 0.985.  That
        growth is a property of bf16 STORAGE, not of the kernels: the float64 oracle with nothing but bf16 rounding of the
        stored tensors deviates from itself by the same amounts (tests/test_oracle.py::
        test_bf16_storage_sensitivity_of_generator_gradients: 0.005 / 0.04-0.08 / 0.06-0.10 / 0.09-0.125).  Limits per
        depth: G.Output* 0.02, G.Block.3 0.10, G.Block.2 0.13, G.Block.1 and G.Input 0.22."""
    seed, b = 21, 64
    S, tr, state = make_trainer(seed, b)
    rng = np.random.default_rng(64)
    z = bf16r(rng.normal(size=(b, 128)))
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy())
    P = T.to_torch(state)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    # ---- generator forward
    with torch.no_grad():
        img = S.Generator(b, labels.cuda(), noise=z.cuda(), groups=2)
        ref_img = T.generator(P, z.to(torch.float64), labels.long(), groups=2)
    diff = (img.to(torch.float64).cpu() - ref_img).abs()
    print("bs64 image max/mean |d|:", diff.max().item(), diff.mean().item())
    assert diff.max().item() < 0.09 and diff.mean().item() < 0.007, (diff.max().item(), diff.mean().item())
    # ---- critic loss, logits and gradients
    loss, _, ref_logits = T.d_loss_fn(P, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    dn = T.trainable_names(P, 'Discriminator')
    ref_g = dict(zip(dn, torch.autograd.grad(loss, [P[k] for k in dn])))
    tr.real_labels.copy_(labels)
    logits = tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
    torch.cuda.synchronize()
    dl = (logits.to(torch.float64).cpu() - ref_logits.detach()).abs()
    print("bs64 logits max |d|:", dl.max().item(), "d_loss", float(tr.d_loss), float(loss))
    assert (dl <= 0.02 * torch.clamp(ref_logits.detach().abs(), min=1.0)).all(), dl.max().item()
    assert abs(float(tr.d_loss) - float(loss)) < 5e-3
    errs = _grad_errors(tr, dn, ref_g)
    print("bs64 D grads (cos, relL2):", {k.split('/', 1)[1]: (round(c, 5), round(l, 4)) for k, (c, l, _, _) in errs.items()})
    bad = []
    for k, (cos, l2, rn, gmax) in errs.items():
        if rn < 1e-9:          # D.Output/b: the hinge gradients of an all-active batch sum to exactly zero
            if gmax > 1e-6:
                bad.append((k, 'abs', gmax))
            continue
        # the label-embedding branch sums gradients that are constant over 16 x 16 pixels and largely cancel
        lim = (0.999, 0.03) if 'mbedding' in k else (0.9999, 0.015)
        if cos < lim[0] or l2 > lim[1]:
            bad.append((k, cos, l2))
    assert not bad, bad
    # ---- generator loss and gradients (critic state as the pass above left it: u advanced once)
    P = T.to_torch(tr.store.state_dict())
    z2 = bf16r(rng.normal(size=(2 * b, 128)))
    fl = torch.tensor(rng.integers(0, 10, 2 * b), dtype=torch.int32)
    loss, _ = T.g_loss_fn(P, z2.to(torch.float64), fl.long())
    gn = T.trainable_names(P, 'Generator')
    ref_g = dict(zip(gn, torch.autograd.grad(loss, [P[k] for k in gn])))
    tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.g_loss) - float(loss)) < 5e-3
    errs = _grad_errors(tr, gn, ref_g)
    print("bs64 G grads (cos, relL2):", {k.split('/', 1)[1]: (round(c, 5), round(l, 4)) for k, (c, l, _, _) in errs.items()})
    bad = []
    for k, (cos, l2, rn, gmax) in errs.items():
        if k.endswith('Biases') and 'G.Output' not in k:
            # a conv bias that feeds a batch norm has an exactly-zero true gradient: absolute bound
            if gmax > 2e-4:
                bad.append((k, 'abs', gmax))
            continue
        lim = 0.02 if 'G.Output' in k else 0.10 if 'G.Block.3' in k else 0.13 if 'G.Block.2' in k else 0.22
        if cos < 0.985 or l2 > lim:
            bad.append((k, cos, l2))
    assert not bad, bad


ZERO_TABLE_TRAVEL_BOUND = 0.5


@pytest.mark.parametrize("stats", ["deterministic", "epilogue_atomics"])
def test_short_training_tracks_the_fp32_restatement(gpu, stats):
    """stats: 'deterministic' = the fixed-order batch-norm statistics (bounds from the 20-seed distribution below);
    'epilogue_atomics' = the PRODUCTION default (statistics summed by the conv epilogues' float atomics, incl. the image-resident
    16x16 kernel's), on the wider bounds run-to-run noise of 0.4 % per update needs (0.6 / 0.8 travel, 0.35 loss windows).

    The only available proxy for "same sample quality as the reference" (no Inception weights, no TensorFlow): 60
    full iterations (300 critic + 60 generator updates) at batch 8 from identical parameters, on IDENTICAL inputs
    (same images, labels, z, fake labels; no dequantisation noise), HIP bf16 trainer vs the fp32 CPU restatement of the
    reference graph.  Individual weights diverge chaotically (TF-Adam with beta1 = 0 turns a sign flip of a ~0 gradient
    into a 2*lr jump), so the comparison is statistical: loss curves averaged over windows, per-tensor parameter
    norms, and the distance travelled from the initial point."""
    from gan_lib_tensorflow_amd import functional as Fn
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    stats_were = Fn.CONV_EPILOGUE_STATS
    Fn.CONV_EPILOGUE_STATS = stats == "epilogue_atomics"
    try:
        _short_training(stats == "deterministic")
    finally:
        Fn.CONV_EPILOGUE_STATS = stats_were


def _short_training(deterministic):
    iters, b = 60, 8
    S, tr, state = make_trainer(31, b)
    P = T.to_torch(state, dtype=torch.float32)
    P0 = {k: v.detach().clone() for k, v in P.items()}
    ot = T.Trainer(P)
    rng = np.random.default_rng(2024)
    hip_d, ref_d, hip_g, ref_g = [], [], [], []
    for it in range(iters):
        if it > 0:
            z2 = bf16r(rng.normal(size=(2 * b, 128)))
            fl = torch.tensor(rng.integers(0, 10, 2 * b), dtype=torch.int32)
            tr._g_forward_backward(z=z2.cuda(), fake_labels=fl.cuda())
            tr._g_apply()
            hip_g.append(float(tr.g_loss))
            ref_g.append(ot.g_step(it, z2.to(torch.float32), fl.long()))
        for _ in range(5):
            z = bf16r(rng.normal(size=(b, 128)))
            labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
            real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
            real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy())
            tr.real_labels.copy_(labels)
            tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
            tr.d_opt.apply()
            hip_d.append(float(tr.d_loss))
            ref_d.append(ot.d_step(it, None, labels.long(), z.to(torch.float32), None, real_pre=real_pre.to(torch.float32)))
        tr.iteration += 1
        tr.iteration_dev.fill_(tr.iteration)
    hip_d, ref_d, hip_g, ref_g = map(np.asarray, (hip_d, ref_d, hip_g, ref_g))
    assert np.isfinite(hip_d).all() and np.isfinite(hip_g).all()
    # first updates: same function of the same inputs (before chaos sets in)
    assert np.abs(hip_d[:5] - ref_d[:5]).max() < 0.05, (hip_d[:5], ref_d[:5])
    # critic loss curve, window means over 50 updates (measured over repeated runs: differences 0.003 .. 0.17 -- the HIP
    # trajectory itself differs from run to run: fp32 atomics order)
    wd = np.abs(hip_d.reshape(-1, 50).mean(1) - ref_d.reshape(-1, 50).mean(1))
    print("short training: d_loss windows", hip_d.reshape(-1, 50).mean(1), ref_d.reshape(-1, 50).mean(1))
    print("short training: g_loss mean/std", hip_g.mean(), hip_g.std(), ref_g.mean(), ref_g.std())
    assert wd.max() < (0.3 if deterministic else 0.35), wd
    # generator loss = -mean(D(G(z))) over 16 samples swings by +-1 from update to update on either trajectory (std ~0.8):
    # only its level over the whole run is comparable
    assert abs(hip_g.mean() - ref_g.mean()) < 0.8 and 0.0 < hip_g.mean() < 5.0, (hip_g.mean(), ref_g.mean())        # measured 0.07 .. 0.34
    # parameter norms and distance travelled, per tensor.  (Conv biases that feed a batch norm are excluded: their true
    # gradient is exactly zero, TF-Adam turns whatever rounding noise arrives into steps of ~lr, and the normalisation
    # removes the bias again -- they random-walk differently on every implementation without touching the function.)
    norms, travels = {}, {}
    for k in ot.g_names + ot.d_names:
        if k.startswith('Generator/') and k.endswith('Biases') and 'G.Output' not in k:
            continue
        a = tr.store.vars[k].detach().float().cpu()
        r = P[k].detach()
        if float(r.norm()) > 1e-3 and float(P0[k].norm()) > 0.5 * float(r.norm()):      # zero-initialised tensors: travel only
            norms[k] = abs(float(a.norm()) / float(r.norm()) - 1.0)
        ta, trf = float((a - P0[k]).norm()), float((r - P0[k]).norm())
        if trf > 1e-3 and r.numel() >= 128:
            travels[k] = abs(ta / trf - 1.0)
    top = lambda d: sorted(d.items(), key=lambda kv: -kv[1])[:3]      # noqa: E731
    print("short training: worst |norm ratio - 1|", top(norms), "worst |travel ratio - 1|", top(travels))
    # measured over repeated runs: norm ratios within 0.003, travel ratios within 0.25 (worst: the zero-initialised G.OutputNorm
    # offset table, whose 60-step random walk is the noisiest)
    # Runs on the deterministic batch-norm statistics (`deterministic_stats`): the trajectory then repeats to 0.02 % per update
    # instead of 0.4 %, and the bounds below are set from the distribution over 20 parameter / data seeds in
    # profiles/r04_travel_ratio_distribution.txt (scratch/travel_dist.py), not from repeated runs of one seed.
    zero_init = lambda k: k.endswith('CondBatchNorm/offset')      # noqa: E731
    assert max(norms.values()) < 0.02, top(norms)
    lim = (0.5, ZERO_TABLE_TRAVEL_BOUND) if deterministic else (0.6, 0.8)
    assert max(v for k, v in travels.items() if not zero_init(k)) < lim[0] and max(v for k, v in travels.items() if zero_init(k)) < lim[1], top(travels)
    # ---- parity at a TRAINED state (spectral norms, conditional-batch-norm tables and Adam-shaped weights have moved):
    # the HIP trainer takes over the restatement's parameters and both differentiate the same losses on the same inputs
    tr.store.load_state_dict({k: v.detach().numpy() for k, v in P.items()})
    tr._refresh_g_prep()
    P64 = T.to_torch({k: v.detach().numpy() for k, v in P.items()})
    z = bf16r(rng.normal(size=(b, 128)))
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    # the trained critic separates the synthetic "real" images from the fakes perfectly (hinge loss exactly 0, zero
    # gradient); feeding it generator samples in the REAL slot keeps the hinge active and the gradients informative
    with torch.no_grad():
        z_r = bf16r(rng.normal(size=(b, 128)))
        real_pre = bf16r(T.generator(P64, z_r.to(torch.float64), labels.long(), groups=2).numpy())
    loss, _, _ = T.d_loss_fn(P64, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    assert float(loss) > 0.5
    dn = T.trainable_names(P64, 'Discriminator')
    ref_gr = dict(zip(dn, torch.autograd.grad(loss, [P64[k] for k in dn])))
    tr.real_labels.copy_(labels)
    tr._d_forward_backward(real_pre=real_pre.cuda(), z=z.cuda())
    torch.cuda.synchronize()
    errs = _grad_errors(tr, dn, ref_gr)
    print("trained-state D loss", float(tr.d_loss), float(loss), "grads (cos, relL2):",
          {k.split('/', 1)[1]: (round(c, 4), round(l, 3)) for k, (c, l, _, _) in errs.items()})
    assert abs(float(tr.d_loss) - float(loss)) < 0.05
    bad = [(k, c, l) for k, (c, l, rn, _) in errs.items() if rn > 1e-9 and (c < 0.999 or l > (0.05 if 'mbedding' in k else 0.03))]    # measured: cosine 0.9999-1.0, L2 0.001-0.017
    assert not bad, bad


def test_trainer_checkpoint_restores_the_optimiser_and_counters(gpu):
    """tf.train.Saver of the reference (:585-590) checkpoints the Adam slots and beta powers with the variables: a
    restored trainer continues the same trajectory (Adam moments, bias-correction step, LR-decay iteration, RNG)."""
    S, tr, _ = make_trainer(41, 8)
    feed = S.synthetic_batches(8, "cuda", seed=3)
    for _ in range(3):
        tr.train_iteration(feed)
    sd = tr.state_dict()
    assert int(sd['Discriminator/adam_t']) == 15 and int(sd['Generator/adam_t']) == 2 and int(sd['_iteration']) == 3
    assert 'Generator/G.Block.1.Conv1/Filters/Adam_1' in sd and sd['Discriminator/D.Output/W/Adam'].shape == (128, 1)
    batches = [next(feed) for _ in range(5)]
    tr.train_iteration(iter(batches))
    torch.cuda.synchronize()
    tr2 = S.SNGANTrainer(batch_size=8, seed=99, use_graphs=False)        # different init: everything comes from the checkpoint
    tr2.load_state_dict(sd)
    assert tr2.iteration == 3 and int(tr2.d_opt.t) == 15 and int(tr2.g_opt.t) == 2 and torch.equal(tr2.rng_state.cpu(), torch.from_numpy(sd['_rng_state']))
    tr2.train_iteration(iter(batches))
    torch.cuda.synchronize()
    for net in ('Generator', 'Discriminator'):
        a, b_ = tr.store.flat[net]["params"], tr2.store.flat[net]["params"]
        d = (a - b_).abs()
        # same inputs, same noise, same Adam state: only fp32 atomics order differs (a fresh Adam would move every weight
        # by ~lr = 2e-4 in its first bias-corrected step)
        assert (d > 5e-5).float().mean().item() < 0.02 and d.mean().item() < 5e-6, (net, (d > 5e-5).float().mean().item(), d.mean().item())
    # a weights-only checkpoint resets the optimiser instead of keeping the old run's moments
    weights_only = {k: v for k, v in sd.items() if not k.endswith(('/Adam', '/Adam_1', '/adam_t')) and not k.startswith('_')}
    tr2.load_state_dict(weights_only)
    assert int(tr2.d_opt.t) == 0 and tr2.iteration == 0 and float(tr2.d_flat['m'].abs().max()) == 0.0 and float(tr2.g_flat['v'].abs().max()) == 0.0


def test_bucketed_generator_update_through_rccl_matches_the_one_piece_update(gpu):
    """Data-parallel generator update (backend "nccl" = RCCL, world_size 1 on this box): forward, FOUR backward segments
    cut at the block boundaries, the bucket of each segment all-reduced on the communication stream while the next
    segment runs, optimiser -- every phase its own hipGraph in one memory pool.  Same seeds, same feed as a trainer that
    runs the update in one piece: same RNG consumption, same loss, parameters equal up to fp32-atomics ordering."""
    # the body runs in tests/rccl_worker.py (a process of its own: an abort in RCCL's teardown must not take the session down --
    # but it FAILS this test: the marker is printed after destroy_process_group() and any non-zero exit code is an error)
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")], env=env,
                       capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:])
    assert "RCCL PATH OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, f"rccl_worker exited with {r.returncode} after its checks and teardown: {r.stderr[-3000:]}"


def test_sngan_critic_acgan_head(gpu):
    """CONDITIONAL and ACGAN (gan_cifar_resnet.py:302-311): the critic grows a second, spectrally normalised 10-way head
    `D.ACGANOutput` on the pooled features; both heads receive gradients."""
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    from gan_lib_tensorflow_amd import functional as Fn
    from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
    S.ACGAN = True
    try:
        store = set_default_store(ParamStore("cuda", seed=3))
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(6, 3072, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
        labels = torch.randint(0, 10, (6,), generator=g, dtype=torch.int32).cuda()
        logits, ac = S.Discriminator(x, labels, update_collection=S.NO_OPS)
        assert logits.shape == (6,) and ac.shape == (6, 10)
        for k in ('Discriminator/D.ACGANOutput/W', 'Discriminator/D.ACGANOutput/b', 'Discriminator/D.ACGANOutput/spectral_norm/u'):
            assert k in store.vars, k
        total = Fn.hinge_g_loss(logits) + 0.1 * Fn.softmax_xent(ac, labels)        # gen_cost + ACGAN_SCALE_G * acgan cost (:476)
        total.backward()
        for k in ('Discriminator/D.ACGANOutput/W', 'Discriminator/D.Output/W', 'Discriminator/D.Block.3.Conv1/Filters'):
            gr = store.vars[k].grad
            assert gr is not None and bool(torch.isfinite(gr).all()) and float(gr.abs().sum()) > 0, k
    finally:
        S.ACGAN = False


def test_tf_saver_checkpoint_round_trip_through_the_trainer(gpu, tmp_path):
    """SNGANTrainer.state_dict() -> the tensors tf.train.Saver holds (TF variable names, <var>/Adam slots, beta powers)
    -> V2 checkpoint files -> optimistic_restore into a differently initialised trainer: same weights, Adam moments
    and step counts (gan_cifar_resnet.py:585-590; common/misc.py:275-307)."""
    from gan_lib_tensorflow_amd.common import tf_checkpoint as C
    S, tr, _ = make_trainer(51, 8)
    feed = S.synthetic_batches(8, "cuda", seed=4)
    for _ in range(2):
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    prefix = str(tmp_path / "model.ckpt-1")
    C.write_checkpoint(prefix, C.checkpoint_from_trainer_state(tr.state_dict()))
    names = {n for n, _, _ in C.list_variables(prefix)}
    assert {'Generator/G.Block.1.Conv1/Filters', 'Generator/G.Block.1.Conv1/Filters/Adam_1', 'Discriminator/D.Output/spectral_norm/u',
            'beta2_power', 'beta2_power_1'} <= names and not any(n.startswith('_') for n in names)
    tr2 = S.SNGANTrainer(batch_size=8, seed=7, use_graphs=False)
    restored = C.optimistic_restore(tr2, prefix)
    assert len(restored) == len(tr.store.vars)
    for k in tr.store.vars:
        assert torch.equal(tr.store.vars[k], tr2.store.vars[k]), k
    assert torch.equal(tr.g_flat['m'], tr2.g_flat['m']) and torch.equal(tr.d_flat['v'], tr2.d_flat['v'])
    assert int(tr2.d_opt.t) == int(tr.d_opt.t) == 10 and int(tr2.g_opt.t) == int(tr.g_opt.t) == 1


def test_run_to_run_noise_of_the_generator_update_is_bounded(gpu):
    """Two executions of the SAME generator update from the SAME state (same noise, same labels) at the headline batch.  The
    conv epilogues add their batch-norm statistics with fp32 atomics (one of 16 partial copies per tower), so the sums arrive
    in a different order on every run, and the generator gradient is ill-conditioned under 16-bit storage
    (tests/test_oracle.py::test_bf16_storage_sensitivity...): measured 1.3-2.4 % relative L2 between two runs of the whole
    gradient buffer.  Bounds: <= 5 % with the epilogue statistics (default), <= 0.3 % with the three-pass statistics
    (GANK_EPILOGUE_STATS=0 / functional.CONV_EPILOGUE_STATS = False: the deterministic-statistics option, measured 0.07 %;
    what remains is the atomics order of the filter-gradient and batch-norm-table sums)."""
    from gan_lib_tensorflow_amd import functional as Fn
    S, tr, _ = make_trainer(23, 64)
    rng = np.random.default_rng(7)
    z2 = bf16r(rng.normal(size=(128, 128))).cuda()
    fl = torch.tensor(rng.integers(0, 10, 128), dtype=torch.int32).cuda()
    were = Fn.CONV_EPILOGUE_STATS
    out = {}
    try:
        for stats in (True, False):
            Fn.CONV_EPILOGUE_STATS = stats
            grads = []
            for _ in range(3):
                tr.g_flat["grads_all"].zero_()
                tr.g_flat["clean"] = True
                tr._g_forward_backward(z=z2, fake_labels=fl)
                torch.cuda.synchronize()
                grads.append(tr.g_flat["grads"].double().clone())
            out[stats] = max(float((g - grads[0]).norm() / grads[0].norm()) for g in grads[1:])
    finally:
        Fn.CONV_EPILOGUE_STATS = were
    print("run-to-run relative L2 of the generator gradient buffer:", out)
    assert out[True] < 0.05 and out[False] < 0.003, out


def test_training_on_a_synthetic_class_conditional_dataset_learns_the_classes(gpu, deterministic_stats):
    """Sample-quality proxy (no Inception weights, no CIFAR): 3 000 iterations of the captured train step at the headline batch
    on ten synthetic classes (tests/synthetic_classes.py), then 200 samples per class through the sampling path.  The generator
    must have learned to condition on the label -- at least 8 of 10 generated class means are nearest to THEIR data class
    mean -- and to land in the data's range (mean RMS distance of the class means below 0.35 in tanh units).
    scratch/long_run.py is the 5 000-iteration form of this, run for bf16 and fp16 + loss scale (profiles/r03_long_run_*)."""
    from tests.synthetic_classes import make_dataset
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    data, labels = make_dataset(200, 0)
    d_dev, l_dev = torch.tensor(data).cuda(), torch.tensor(labels).cuda()
    tr = S.SNGANTrainer(batch_size=64, seed=0)
    g = torch.Generator(device='cpu').manual_seed(1)

    def batches():
        while True:
            idx = torch.randperm(len(labels), generator=g)[:64].cuda()
            yield d_dev[idx].contiguous(), l_dev[idx].contiguous()
    feed = batches()
    for _ in range(3000):
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.use_graphs and bool(torch.isfinite(tr.g_flat["params"]).all()) and bool(torch.isfinite(tr.d_flat["params"]).all())
    real = (2.0 * (data.astype(np.float64) / 256.0 - 0.5)).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1).reshape(-1, 3072)
    rmeans = np.stack([real[labels == c].mean(0) for c in range(10)])
    means = np.zeros((10, 3072))
    for c in range(10):
        lab = torch.full((100,), c, dtype=torch.int32, device='cuda')
        means[c] = np.concatenate([tr.sample(100, labels=lab).float().cpu().numpy() for _ in range(2)]).mean(0)
    d = np.sqrt(((means[:, None, :] - rmeans[None, :, :]) ** 2).mean(2))
    sep = np.sqrt(((rmeans[:, None, :] - rmeans[None, :, :]) ** 2).mean(2))
    print("class-mean RMS error:", np.round(np.diag(d), 3), "nearest:", d.argmin(1), "smallest class separation:", round(float(sep[sep > 0].min()), 3))
    # measured: 9 / 10 classes recovered, mean RMS 0.28 (the data's classes are 0.30 .. 0.72 apart; 10 / 10 and 0.23 after 5 000 iterations)
    assert (d.argmin(1) == np.arange(10)).sum() >= 8, d.argmin(1)
    assert float(np.diag(d).mean()) < 0.35, np.diag(d).mean()


def test_dev_loss_is_forward_only_and_advances_u(gpu):
    """The dev-loss evaluation (gan_cifar_resnet.py:639-647): disc_cost forward only, critic with update_collection=None -- the
    spectral-norm u vectors advance on every evaluation (sn.py:48-56), weights / Adam state / gradient buffers do not move, and
    the value is the oracle's disc_cost for the same inputs."""
    b = 8
    S, tr, state = make_trainer(61, b)
    rng = np.random.default_rng(5)
    z = bf16r(rng.normal(size=(b, 128)))
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy())
    P = T.to_torch(state)
    ref, new_u, _ = T.d_loss_fn(P, None, labels.long(), z.to(torch.float64), None, real_pre=real_pre.to(torch.float64))
    u_name = 'Discriminator/D.Block.2.Conv1/filters/spectral_norm/u'
    u0 = tr.store.vars[u_name].clone()
    p0, m0 = tr.d_flat["params"].clone(), tr.d_flat["m"].clone()
    tr.d_flat["grads_all"].zero_()
    tr.d_flat["clean"] = True
    got = tr.dev_disc_cost(None, labels, z=z.cuda(), real_pre=real_pre.cuda())
    assert abs(got - float(ref)) < 0.02, (got, float(ref))
    u1 = tr.store.vars[u_name].clone()
    assert float((u1 - u0).abs().max()) > 1e-3                                           # u <- u_final
    assert float((u1.double().cpu().reshape(-1) - new_u[u_name].reshape(-1)).abs().max()) < 1e-4
    assert torch.equal(tr.d_flat["params"], p0) and torch.equal(tr.d_flat["m"], m0) and int(tr.d_opt.t) == 0
    assert float(tr.d_flat["grads_all"].abs().max()) == 0.0 and tr.d_flat["clean"] is True
    tr.dev_disc_cost(None, labels, z=z.cuda(), real_pre=real_pre.cuda())
    assert float((tr.store.vars[u_name] - u1).abs().max()) > 0                           # ... and again on the next evaluation
    # through the uint8 feed and an epoch of batches
    feed = [(real_u8.numpy(), labels.numpy()), (real_u8.numpy(), labels.numpy())]
    assert np.isfinite(tr.dev_loss(feed))
    # the training step still replays after an evaluation
    batches = S.synthetic_batches(b, "cuda", seed=2)
    for _ in range(3):
        tr.train_iteration(batches)
    assert bool(torch.isfinite(tr.d_flat["params"]).all())


def test_get_loss_least_squares_and_sigmoid_branches(gpu):
    """common/misc.py:353-394: LSGAN, CGAN, Modified_MiniMax, MiniMax -- values and d loss / d logits against float64."""
    from gan_lib_tensorflow_amd.common.misc import get_loss
    rng = np.random.default_rng(9)
    real = bf16r(rng.normal(size=24) * 3)
    fake = bf16r(rng.normal(size=40) * 3)

    def ref(kind, r, f):
        sp = torch.nn.functional.softplus
        if kind == 'LSGAN':
            return (((1 - r) ** 2).mean() + (f ** 2).mean()) / 2, ((1 - f) ** 2).mean() / 2
        d = sp(-r).mean() + sp(f).mean()
        return d, (sp(-f).mean() if kind in ('CGAN', 'Modified_MiniMax') else -sp(f).mean())
    for kind in ('LSGAN', 'CGAN', 'Modified_MiniMax', 'MiniMax'):
        r = real.cuda().requires_grad_(True)
        f = fake.cuda().requires_grad_(True)
        d_loss, g_loss = get_loss(r, f, kind)
        rr = real.double().requires_grad_(True)
        ff = fake.double().requires_grad_(True)
        d_ref, g_ref = ref(kind, rr, ff)
        assert abs(float(d_loss) - float(d_ref)) < 1e-5 and abs(float(g_loss) - float(g_ref)) < 1e-5, kind
        d_loss.backward()
        gr, gf = torch.autograd.grad(d_ref, [rr, ff])
        assert float((r.grad.double().cpu() - gr).abs().max()) < 1e-2 * float(gr.abs().max()) + 1e-6, kind      # the gradient leaves the loss launch in bf16
        assert float((f.grad.double().cpu() - gf).abs().max()) < 1e-2 * float(gf.abs().max()) + 1e-6, kind
        f2 = fake.cuda().requires_grad_(True)
        _, g_loss = get_loss(real.cuda(), f2, kind)
        g_loss.backward()
        (gg,) = torch.autograd.grad(g_ref, [ff])
        assert float((f2.grad.double().cpu() - gg).abs().max()) < 1e-2 * float(gg.abs().max()) + 1e-6, kind


def test_image_side_filter_gradients_inside_the_convmeanpool_input_gradient(gpu, monkeypatch):
    """functional.FUSE_IMAGE_WGRAD (round 5): in a critic update D.Block.1.Conv1's and D.Block.1.Shortcut's filter / bias gradients
    come out of the ConvMeanPool input-gradient launch (gank_cpool_res_dgrad_image_wgrad; the 33.5-MB tensor between them is never
    stored).  Same inputs, same state: the fused launch IS taken (once per critic pass), every critic gradient equals the
    two-launch path's up to summation order (relative L2 <= 2e-3 per tensor; the four tensors concerned are printed), the loss is
    identical; the generator update -- where the image needs a gradient -- does not take it."""
    from gan_lib_tensorflow_amd import functional as Fn
    from gan_lib_tensorflow_amd import kernels as K
    seed, b = 33, 16
    S, tr, state = make_trainer(seed, b)
    rng = np.random.default_rng(12)
    z = bf16r(rng.normal(size=(b, 128))).cuda()
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy()).cuda()
    tr.real_labels.copy_(labels)
    with torch.no_grad():       # ONE set of fakes for both passes (the generator's epilogue statistics are summed by atomics: two passes differ in the last bit)
        fake = S.Generator(b, tr.real_labels, noise=z, groups=2)
    calls = []
    orig = K.cpool_res_dgrad_image_wgrad
    monkeypatch.setattr(K, "cpool_res_dgrad_image_wgrad", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    u0 = {k: v.clone() for k, v in tr.store.vars.items() if k.endswith('spectral_norm/u')}
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(Fn, "FUSE_IMAGE_WGRAD", fused)
        for k, v in u0.items():
            tr.store.vars[k].copy_(v)                  # update_collection=None advanced u: both passes start from the same state
        calls.clear()
        tr._d_forward_backward(real_pre=real_pre, fake=fake)
        torch.cuda.synchronize()
        assert len(calls) == (1 if fused else 0), (fused, len(calls))
        out[fused] = (float(tr.d_loss), tr.d_flat["grads"].double().clone())
    assert out[True][0] == out[False][0]
    names = tr.d_flat['names']
    worst = {}
    for k in names:
        o, nel = tr.d_flat['offsets'][k], tr.store.vars[k].numel()
        a, r = out[True][1][o:o + nel], out[False][1][o:o + nel]
        if float(r.norm()) > 1e-12:
            worst[k] = float((a - r).norm() / r.norm())
    print("fused vs two-launch image-side gradients:", {k.split('/', 1)[1]: f"{v:.1e}" for k, v in worst.items() if 'D.Block.1.Conv1' in k or 'D.Block.1.Shortcut' in k})
    others = {k: v for k, v in worst.items() if 'D.Block.1.Conv1' not in k and 'D.Block.1.Shortcut' not in k}
    assert max(others.values()) < 1e-4, sorted(others.items(), key=lambda kv: -kv[1])[:4]            # everything else: the same launches, atomics order only
    assert max(worst.values()) < 2e-3, sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    monkeypatch.setattr(Fn, "FUSE_IMAGE_WGRAD", True)
    calls.clear()
    tr._g_forward_backward()
    torch.cuda.synchronize()
    assert len(calls) == 0 and bool(torch.isfinite(tr.g_flat["grads"]).all()) and float(tr.g_flat["grads"].abs().sum()) > 0


def test_launch_folds_of_the_critic_update_match_the_separate_launches(gpu, monkeypatch):
    """Late round 5: five launches of a critic update became extra workgroups of their neighbours (functional.POOLED_LABEL_PART +
    LABEL_BWD_IN_SUM_SLABS, SHORTCUT_IN_TABLE_LAUNCH, CONV1X1_BWD_ONE_LAUNCH, IMAGE_CONV_PAIR; gan_cifar_resnet.py:172-184,212-234,276-284 forward and backward).
    Same inputs, same state, headline batch: every fold IS taken once per critic pass, the loss agrees to the last bf16 digit of the
    logits and every critic gradient with the separate launches' up to rounding (two of the folds change a summation order in front of
    a bf16 rounding: relative L2 <= 1e-3 per tensor, the worst ones are printed)."""
    from gan_lib_tensorflow_amd import functional as Fn
    from gan_lib_tensorflow_amd import kernels as K
    seed, b = 35, 64
    S, tr, state = make_trainer(seed, b)
    rng = np.random.default_rng(14)
    z = bf16r(rng.normal(size=(b, 128))).cuda()
    labels = torch.tensor(rng.integers(0, 10, b), dtype=torch.int32)
    real_u8 = torch.tensor(rng.integers(0, 256, (b, 3072)), dtype=torch.uint8)
    real_pre = bf16r(T.preprocess_real(real_u8, torch.zeros(b, 3072, dtype=torch.float64), torch.float64).numpy()).cuda()
    tr.real_labels.copy_(labels)
    with torch.no_grad():
        fake = S.Generator(b, tr.real_labels, noise=z, groups=2)
    calls = {"conv1x1_wgrad_dgrad": 0, "image_conv_pair_fprop": 0, "sum_slabs_with_label_gradients": 0, "table_shortcut": 0}
    orig_s = K.sum_slabs
    monkeypatch.setattr(K, "sum_slabs", lambda jobs, label=None: (calls.__setitem__("sum_slabs_with_label_gradients", calls["sum_slabs_with_label_gradients"] + (1 if label is not None else 0)),
                                                                  orig_s(jobs, label))[1])
    for name in ("conv1x1_wgrad_dgrad", "image_conv_pair_fprop"):
        orig = getattr(K, name)
        monkeypatch.setattr(K, name, (lambda o, nm: lambda *a, **k: (calls.__setitem__(nm, calls[nm] + 1), o(*a, **k))[1])(orig, name))
    orig_t = K.label_conv3x3_table_pooled
    monkeypatch.setattr(K, "label_conv3x3_table_pooled",
                        lambda *a, **k: (calls.__setitem__("table_shortcut", calls["table_shortcut"] + (1 if (len(a) > 6 and a[6] is not None) else 0)), orig_t(*a, **k))[1])
    u0 = {k: v.clone() for k, v in tr.store.vars.items() if k.endswith('spectral_norm/u')}
    out = {}
    for folded in (True, False):
        for sw in ("POOLED_LABEL_PART", "LABEL_BWD_IN_SUM_SLABS", "SHORTCUT_IN_TABLE_LAUNCH", "CONV1X1_BWD_ONE_LAUNCH", "IMAGE_CONV_PAIR"):
            monkeypatch.setattr(Fn, sw, folded)
        for k, v in u0.items():
            tr.store.vars[k].copy_(v)
        for k in calls:
            calls[k] = 0
        tr._d_forward_backward(real_pre=real_pre, fake=fake)
        torch.cuda.synchronize()
        assert all(v == (1 if folded else 0) for v in calls.values()), (folded, calls)
        out[folded] = (float(tr.d_loss), tr.d_flat["grads"].double().clone())
    assert abs(out[True][0] - out[False][0]) <= 2e-3 * max(1.0, abs(out[False][0])), (out[True][0], out[False][0])
    worst = {}
    for k in tr.d_flat['names']:
        o, nel = tr.d_flat['offsets'][k], tr.store.vars[k].numel()
        a, r = out[True][1][o:o + nel], out[False][1][o:o + nel]
        if float(r.norm()) > 1e-12:
            worst[k] = float((a - r).norm() / r.norm())
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("folded vs separate launches, worst critic gradients:", [(k.split('/', 1)[1], f"{v:.1e}") for k, v in top])
    assert top[0][1] < 1e-3, top          # (measured: 2.5e-7 -- the changed summation orders rarely cross a bf16 rounding boundary)
