"""PGGAN configuration (BASELINE.json config 4; PGGAN/model_nvidia.py, PGGAN/train.py) on a real MI355X against the float64
oracle (oracle/ref_pggan.py): the small operators of the path (fade-in blend, minibatch-std, legacy bilinear resize), the
generator and the critic at two stages of the progression (with and without a block fading in), both losses with their
gradients, the spectral-norm `u` policy of the two critic passes, and training steps.  bf16 activations / fp32 accumulate;
tolerances stated at each assertion."""
import numpy as np
import pytest
import torch

from oracle import ref_pggan as G
from oracle import ref_torch as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gan_lib_tensorflow_amd import kernels
    kernels.lib()
    return torch.device("cuda")


def bf(a):
    t = torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)
    return t.to(torch.float64), t.cuda().contiguous()


def rel(got, ref):
    got = got.detach().to(torch.float64).cpu()
    ref = torch.as_tensor(ref, dtype=torch.float64)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-300))


def l2(got, ref):
    got, ref = got.detach().to(torch.float64).cpu().flatten(), ref.detach().to(torch.float64).flatten()
    assert torch.isfinite(got).all()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-300))


def cos(got, ref):
    got, ref = got.detach().to(torch.float64).cpu().flatten(), ref.detach().to(torch.float64).flatten()
    return float((got @ ref) / max(float(got.norm() * ref.norm()), 1e-300))


def test_blend_minibatch_std_resize(gpu):
    from gan_lib_tensorflow_amd import functional as Fn, kernels as K
    rng = np.random.default_rng(0)
    a, at = bf(rng.normal(size=(6, 8, 8, 64)))
    b, bt = bf(rng.normal(size=(6, 8, 8, 64)))
    at.requires_grad_(True); bt.requires_grad_(True)
    y = Fn.blend(at, bt, 0.3)
    assert rel(y, 0.7 * a + 0.3 * b) < 1e-2
    g, gt = bf(rng.normal(size=y.shape))
    y.backward(gt)
    assert rel(at.grad, 0.7 * g) < 1e-2 and rel(bt.grad, 0.3 * g) < 1e-2
    # minibatch std (model_nvidia.py:20-29): value and gradient against autograd of the restatement
    for shape in ((8, 4, 4, 512), (5, 8, 8, 24)):
        x, xt = bf(rng.normal(size=shape) * 1.5 + 0.2)
        xr = x.clone().requires_grad_(True)
        ref = G.minibatch_std(xr)
        xt.requires_grad_(True)
        out = Fn.minibatch_std(xt)
        assert out.shape == ref.shape and rel(out, ref.detach()) < 1e-2
        assert np.allclose(ref.detach().numpy(), G.minibatch_std_numpy(x.numpy()))
        dy, dyt = bf(rng.normal(size=ref.shape))
        ref.backward(dy)
        out.backward(dyt)
        assert l2(xt.grad, xr.grad) < 1e-2
    # tf.image.resize_images (TF 1.5 bilinear, align_corners=False): down and up, as train.py:88-92 chains them
    x, xt = bf(rng.uniform(-1, 1, size=(4, 32, 32, 3)))
    for size in ((16, 16), (8, 8), (4, 4), (64, 64)):
        assert rel(K.resize_bilinear(xt, size), G.resize_bilinear(x, size)) < 1e-2
    half = K.resize_bilinear(xt, (8, 8))
    assert rel(K.resize_bilinear(half, (16, 16)), G.resize_bilinear(half.double().cpu(), (16, 16))) < 1e-2
    known = G.resize_bilinear(np.arange(16.).reshape(1, 4, 4, 1), (8, 8))[0, :, :, 0]
    assert known[0, 1] == 0.5 and known[1, 0] == 2.0 and known[7, 7] == 15.0          # src = dst / 2, last row / column clamped


def make(bc, trans, batch, seed=3):
    from gan_lib_tensorflow_amd.PGGAN.train import PGGANTrainer, default_args
    args = default_args(batch_size=batch, block_count=bc, image_size=4 * 2 ** bc, trans=trans, max_iter=1000)
    tr = PGGANTrainer(args, seed=seed)
    return tr, tr.store.state_dict()


def _bad_grads(tr, names, gref, l2_max=0.1):
    """tensors outside cosine >= 0.99 / relative L2 <= l2_max; a gradient that is ~0 in the restatement (the last layer's bias
    when every hinge margin is active: -1/n per real + 1/n per fake) is bounded absolutely instead"""
    bad = []
    gmax = max(float(g.abs().max()) for g in gref.values())
    for k in names:
        g, r = tr.store.vars[k].main_grad, gref[k]
        if float(r.abs().max()) < 1e-6 * gmax:
            if float(g.abs().max()) > 1e-3 * gmax:
                bad.append((k, 'abs', float(g.abs().max())))
            continue
        c, e = cos(g, r), l2(g, r)
        if c < 0.99 or e > l2_max:
            bad.append((k, c, e))
    return bad


@pytest.mark.parametrize("bc,trans", [(1, True), (2, False), (3, True)])
def test_pggan_model_losses_gradients_vs_oracle(gpu, bc, trans):
    """Generator images and critic logits, the two losses and their gradients w.r.t. every trainable variable, and the `u`
    vectors after a critic update's two passes (real: written; fake: NO_OPS), against the float64 restatement from the same
    parameters, noise and fade-in weight.  Bounds: images |delta| <= 2e-2 of the range, logits <= 3e-2 * max(1, |ref|),
    gradients cosine >= 0.99 and relative L2 <= 0.1 per tensor (0.15 at the 32x32 stage; no batch statistics on this path: bf16
    storage is the only error)."""
    batch, alpha = 8, 0.37
    tr, state = make(bc, trans, batch)
    names = sorted(state)
    assert names == sorted(G.init_params(0, bc, trans)), set(names) ^ set(G.init_params(0, bc, trans))      # the reference's variable names
    P = T.to_torch(state)
    rng = np.random.default_rng(bc)
    z, zt = bf(rng.normal(size=(batch, 512)))
    size = 4 * 2 ** bc
    real, realt = bf(np.clip(rng.normal(size=(batch, size, size, 3)) * 0.5, -1, 1))
    with torch.no_grad():
        img = tr.model.get_generator(zt, alpha, reuse=True)
        lg = tr.model.get_discriminator(realt, alpha, update_collection='NO_OPS', reuse=True)
        img_ref = G.generator(P, z, alpha, bc, trans)
        lg_ref, _ = G.discriminator(P, real, alpha, bc, trans)
    assert img.shape == (batch, size, size, 3)
    scale = max(1.0, float(img_ref.abs().max()))
    assert float((img.double().cpu() - img_ref).abs().max()) < 2e-2 * scale
    assert float((lg.double().cpu() - lg_ref).abs().max()) < 3e-2 * max(1.0, float(lg_ref.abs().max()))
    # critic loss: values, gradients, u policy
    loss_ref, new_u = G.d_loss(P, real, z, alpha, bc, trans)
    dn = T.trainable_names(P, 'd_net')
    gref = dict(zip(dn, torch.autograd.grad(loss_ref, [P[k] for k in dn])))
    loss = tr.d_loss(realt, z=zt, alpha=alpha)
    tr._backward(loss)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-2 * max(1.0, abs(float(loss_ref)))
    l2_max = 0.1 if bc <= 2 else 0.15       # three up + four down blocks deep: measured 0.125 at g_net/G.Input/W (cosine 0.992)
    assert _bad_grads(tr, dn, gref, l2_max) == []
    for k, u in new_u.items():
        assert rel(tr.store.vars[k], u) < 1e-3, k                   # written once, by the real pass
    tr.d_flat['grads'].zero_()
    # generator loss
    P = T.to_torch(tr.store.state_dict())                            # u has advanced
    loss_ref = G.g_loss(P, z, alpha, bc, trans)
    gn = T.trainable_names(P, 'g_net')
    gref = dict(zip(gn, torch.autograd.grad(loss_ref, [P[k] for k in gn])))
    loss = tr.g_loss(z=zt, alpha=alpha)
    tr._backward(loss)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-2 * max(1.0, abs(float(loss_ref)))
    assert _bad_grads(tr, gn, gref, l2_max) == []
    assert all(float(tr.store.vars[k].main_grad.abs().max()) == 0.0 for k in dn)     # gen_cost moves g_vars only (train.py:132)


def test_pggan_full_resolution_forward_vs_oracle(gpu):
    """BASELINE.json config 4 at its last stage: block_count 6 = 256 x 256 (channels 512,512,512,256,128,64), generator images
    and critic logits against the float64 restatement (forward only: the float64 backward pass of this size takes minutes)."""
    bc, trans, batch, alpha = 6, True, 2, 0.6
    tr, state = make(bc, trans, batch, seed=9)
    P = T.to_torch(state, requires_grad=False)
    rng = np.random.default_rng(6)
    z, zt = bf(rng.normal(size=(batch, 512)))
    with torch.no_grad():
        img = tr.model.get_generator(zt, alpha, reuse=True)
        img_ref = G.generator(P, z, alpha, bc, trans)
        lg = tr.model.get_discriminator(img, alpha, update_collection='NO_OPS', reuse=True)
        lg_ref, _ = G.discriminator(P, img.double().cpu(), alpha, bc, trans)
    assert img.shape == (batch, 256, 256, 3)
    scale = max(1.0, float(img_ref.abs().max()))
    d = (img.double().cpu() - img_ref).abs()
    print("pggan 256x256 image max |delta|", float(d.max()), "mean", float(d.mean()), "range", scale)
    assert float(d.max()) < 3e-2 * scale and float(d.mean()) < 3e-3 * scale
    assert float((lg.double().cpu() - lg_ref).abs().max()) < 3e-2 * max(1.0, float(lg_ref.abs().max()))


def test_pggan_training_steps(gpu, deterministic_stats):
    """train.py:185-193 at the 16x16 stage with a block fading in: 1 generator + 5 critic updates per step, alpha = step /
    max_iter, real rows resized 32 -> 8 -> 16; parameters move by at most ~lr per update and stay finite."""
    from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches
    tr, _ = make(2, True, 16, seed=5)
    feed = synthetic_batches(16, "cuda", seed=2)
    x = tr.real_images(next(feed)[0])
    assert x.shape == (16, 16, 16, 3) and float(x.abs().max()) <= 1.01
    p0d, p0g = tr.d_flat['params'].clone(), tr.g_flat['params'].clone()
    for _ in range(3):
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.step == 3 and int(tr.d_opt['t']) == 15 and int(tr.g_opt['t']) == 3 and abs(tr.alpha() - 0.003) < 1e-12
    for flat, p0, n in ((tr.d_flat, p0d, 15), (tr.g_flat, p0g, 3)):
        assert bool(torch.isfinite(flat['params']).all()) and float(flat['grads'].abs().max()) == 0.0     # cleared by the Adam launch
        moved = (flat['params'] - p0).abs()
        assert 1e-5 < float(moved.max()) < n * 1e-4 * 1.5
    assert all(np.isfinite(float(v)) for v in tr.losses.values())
    img = tr.sample(10)
    assert img.shape == (10, 16, 16, 3) and bool(torch.isfinite(img.float()).all())


@pytest.mark.parametrize("bc,trans,res", [(4, True, 64), (6, False, 256)])
def test_pggan_training_steps_at_64_fading_and_256(gpu, bc, trans, res, deterministic_stats):
    """BASELINE.json config 4 at its later stages (batch 16): 64 x 64 with the newest block fading in and the final 256 x 256
    stage -- two captured train steps each; parameters move by at most ~lr per update, stay finite, gradients are cleared by the
    optimiser launch, samples have the stage's resolution."""
    from gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet import synthetic_batches
    tr, _ = make(bc, trans, 16, seed=5)
    feed = synthetic_batches(16, "cuda", seed=2)
    x = tr.real_images(next(feed)[0])
    assert x.shape == (16, res, res, 3) and float(x.abs().max()) <= 1.01
    p0d, p0g = tr.d_flat['params'].clone(), tr.g_flat['params'].clone()
    for _ in range(2):
        tr.train_iteration(feed)
    torch.cuda.synchronize()
    assert tr.step == 2 and int(tr.d_opt['t']) == 10 and int(tr.g_opt['t']) == 2
    for flat, p0, n in ((tr.d_flat, p0d, 10), (tr.g_flat, p0g, 2)):
        assert bool(torch.isfinite(flat['params']).all()) and float(flat['grads'].abs().max()) == 0.0
        moved = (flat['params'] - p0).abs()
        assert 1e-5 < float(moved.max()) < n * 1e-4 * 3      # beta1 = 0: a step is lr * g / rms(g history), above lr when a gradient outgrows its history
    assert all(np.isfinite(float(v)) for v in tr.losses.values())
    img = tr.sample(4)
    assert img.shape == (4, res, res, 3) and bool(torch.isfinite(img.float()).all())
