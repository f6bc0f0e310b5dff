"""CPU tests of the oracle itself: NumPy float64 (hand-derived gradients) vs torch-CPU float64
autograd, known-answer tests, finite differences.  No GPU, no product code."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as R
from oracle import ref_torch as T



def t64(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float64)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("k,ci,co,h", [(3, 5, 4, 6), (1, 3, 7, 4), (3, 3, 2, 5), (4, 2, 3, 6)])
def test_conv_numpy_vs_torch_and_loops(k, ci, co, h):
    rng = np.random.default_rng(k * 100 + ci)
    x = rng.normal(size=(2, h, h, ci))
    w = rng.normal(size=(k, k, ci, co))
    b = rng.normal(size=co)
    dy = rng.normal(size=(2, h, h, co))
    y = R.conv2d_same(x, w, b)
    np.testing.assert_allclose(y, R.conv2d_direct_loops(x, w, b), atol=1e-12)
    xt, wt, bt = t64(x, True), t64(w, True), t64(b, True)
    yt = T.conv2d_same(xt, wt, bt)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-11)
    yt.backward(t64(dy))
    dx, dw, db = R.conv2d_same_grads(x, w, dy)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(dw, wt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-11)


def test_conv_delta_kernel_is_identity():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(1, 5, 5, 3))
    w = np.zeros((3, 3, 3, 3))
    for c in range(3):
        w[1, 1, c, c] = 1.
    np.testing.assert_allclose(R.conv2d_same(x, w), x, atol=0)


def test_conv_stride2_same_vs_torch():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 8, 8, 3))
    w = rng.normal(size=(4, 4, 3, 5))
    y = R.conv2d_same(x, w, None, stride=2)
    # TF SAME for k=4,s=2 on 8: pad 1 before, 1 after
    xt = torch.nn.functional.pad(t64(x).permute(0, 3, 1, 2), (1, 1, 1, 1))
    yt = torch.nn.functional.conv2d(xt, t64(w).permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.numpy(), atol=1e-11)


@pytest.mark.parametrize("k", [3, 4, 5])
def test_deconv_vs_torch_conv_transpose(k):
    """Deconv2D (deconv2d.py:99-109): conv2d_transpose stride 2 SAME = autograd dgrad of the stride-2 conv."""
    rng = np.random.default_rng(k)
    x = rng.normal(size=(2, 4, 4, 3))            # [N,H,W,Cin]
    f = rng.normal(size=(k, k, 5, 3))            # [k,k,Cout,Cin]
    b = rng.normal(size=5)
    y = R.deconv2d_same(x, f, b)
    assert y.shape == (2, 8, 8, 5)
    # independent: gradient of conv2d_same(stride 2) wrt its input, with upstream = x
    inp = t64(np.zeros((2, 8, 8, 5)), True)
    out = T_conv_stride(inp, t64(f), 2)
    out.backward(t64(x))
    np.testing.assert_allclose(y - b, inp.grad.numpy(), atol=1e-11)
    # grads of the deconv itself by autograd through the same construction
    dy = rng.normal(size=y.shape)
    dx, df, db = R.deconv2d_same_grads(x, f, dy)
    xt, ft = t64(x, True), t64(f, True)
    inp = t64(np.zeros((2, 8, 8, 5)), True)
    out = T_conv_stride(inp, ft, 2)
    (yt,) = torch.autograd.grad(out, inp, xt, create_graph=True)
    yt.backward(t64(dy))
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(df, ft.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(db, dy.sum(axis=(0, 1, 2)), atol=1e-11)


def T_conv_stride(x, w, stride):
    k = w.shape[0]
    h = x.shape[1]
    _, pt, pb = R.same_pads(h, k, stride)
    xn = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pt, pb, pt, pb))
    return torch.nn.functional.conv2d(xn, w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)


def test_upsample_is_depth_to_space_of_concat_and_pool_inverts_it():
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, 3, 4, 5))
    up = R.upsample_nn2x(x)
    for i in range(2):
        for j in range(2):
            np.testing.assert_array_equal(up[:, i::2, j::2, :], x)
    np.testing.assert_allclose(R.meanpool2x2(up), x, atol=1e-15)
    np.testing.assert_array_equal(up, T.upsample_nn2x(t64(x)).numpy())
    dy = rng.normal(size=up.shape)
    np.testing.assert_allclose(R.upsample_nn2x_grad(dy), 4 * R.meanpool2x2(dy), atol=1e-14)
    np.testing.assert_allclose(R.meanpool2x2_grad(x), R.upsample_nn2x(x) / 4, atol=0)


def test_1x1_conv_commutes_with_pool_and_upsample():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 4, 4, 6))
    w = rng.normal(size=(1, 1, 6, 3))
    b = rng.normal(size=3)
    np.testing.assert_allclose(R.meanpool2x2(R.conv2d_same(x, w, b)), R.conv2d_same(R.meanpool2x2(x), w, b), atol=1e-12)
    np.testing.assert_allclose(R.upsample_nn2x(R.conv2d_same(x, w, b)), R.conv2d_same(R.upsample_nn2x(x), w, b), atol=1e-12)


@pytest.mark.parametrize("K,C", [(27, 8), (12, 1), (3, 16), (40, 40)])
def test_sn_forward_backward_vs_autograd(K, C):
    rng = np.random.default_rng(K + C)
    W = rng.normal(size=(K, C))
    u = T.trunc_normal(rng, (1, C)).astype(np.float64)
    G = rng.normal(size=(K, C))
    Wb, u1, sigma, v = R.sn_forward(W, u)
    Wt = t64(W, True)
    Wbt, u1t, sigt = T.spectral_normed_weight(Wt, t64(u))
    np.testing.assert_allclose(Wb, Wbt.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(u1, u1t.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(sigma, float(sigt.detach()), rtol=1e-12)
    Wbt.backward(t64(G))
    dW = R.sn_backward(W, u, G)
    np.testing.assert_allclose(dW, Wt.grad.numpy(), rtol=1e-9, atol=1e-12)
    # the stop-gradient (Chainer-style) variant is a DIFFERENT gradient when u is not converged
    dW_sg = G / sigma - (np.sum(G * W) / sigma ** 2) * np.outer(v.ravel(), u1.ravel())
    if C > 1:
        assert np.abs(dW_sg - dW).max() > 1e-4


def test_sn_rank1_sigma_exact_and_converged_top_singular_value_is_one():
    rng = np.random.default_rng(5)
    p, q = rng.normal(size=(7, 1)), rng.normal(size=(1, 4))
    W = p @ q
    _, _, sigma, _ = R.sn_forward(W, rng.normal(size=(1, 4)))
    np.testing.assert_allclose(sigma, np.linalg.norm(p) * np.linalg.norm(q), rtol=1e-12)
    W = rng.normal(size=(9, 5))
    u = rng.normal(size=(1, 5))
    for _ in range(500):
        _, u, sigma, _ = R.sn_forward(W, u)
    Wb, _, _, _ = R.sn_forward(W, u)
    np.testing.assert_allclose(np.linalg.svd(Wb, compute_uv=False)[0], 1., rtol=1e-9)


def test_sn_conv_filter_reshape_is_rows_of_kh_kw_cin():
    rng = np.random.default_rng(6)
    W = rng.normal(size=(3, 3, 2, 4))
    u = rng.normal(size=(1, 4))
    Wb, _, sigma, _ = R.sn_forward(W, u)
    Wb2, _, sigma2, _ = R.sn_forward(W.reshape(18, 4), u)
    np.testing.assert_allclose(Wb.reshape(18, 4), Wb2)


@pytest.mark.parametrize("groups", [1, 2])
def test_cbn_forward_backward_vs_autograd(groups):
    rng = np.random.default_rng(7)
    x = rng.normal(size=(4, 3, 3, 5)) * 2 + 1
    labels = np.array([1, 0, 1, 2])
    gamma = rng.normal(size=(3, 5))
    beta = rng.normal(size=(3, 5))
    dy = rng.normal(size=x.shape)
    y, cache = R.cond_batchnorm_forward(x, labels, gamma, beta, groups)
    xt, gt, bt = t64(x, True), t64(gamma, True), t64(beta, True)
    yt = T.cond_batchnorm(xt, torch.tensor(labels), gt, bt, groups)
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-12)
    yt.backward(t64(dy))
    dx, dg, db = R.cond_batchnorm_backward(dy, labels, gamma, cache, groups)
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dg, gt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-10)


def test_cbn_unit_gamma_gives_zero_mean_unit_var():
    rng = np.random.default_rng(8)
    x = rng.normal(size=(8, 4, 4, 3)) * 3 + 2
    y, _ = R.cond_batchnorm_forward(x, np.zeros(8, int), np.ones((1, 3)), np.zeros((1, 3)))
    np.testing.assert_allclose(y.mean(axis=(0, 1, 2)), 0, atol=1e-12)
    np.testing.assert_allclose(y.var(axis=(0, 1, 2)), 1, atol=1e-4)   # eps=1e-5 bias


def test_hinge_known_answers_and_grads():
    loss, d = R.hinge_d_loss(np.array([1., 1., -1., -1.]), 2)
    assert loss == 0. and not d.any()
    loss, d = R.hinge_d_loss(np.array([0., 2., 0., -3.]), 2)
    np.testing.assert_allclose(loss, 0.5 + 0.5)
    np.testing.assert_allclose(d, [-0.5, 0, 0.5, 0])
    loss, d = R.hinge_g_loss(np.array([1., 3.]))
    np.testing.assert_allclose(loss, -2.)
    np.testing.assert_allclose(d, [-0.5, -0.5])
    lg = np.random.default_rng(9).normal(size=(5, 10))
    lb = np.array([0, 3, 9, 3, 1])
    loss, d = R.softmax_xent(lg, lb)
    lt = t64(lg, True)
    l2 = torch.nn.functional.cross_entropy(lt, torch.tensor(lb))
    l2.backward()
    np.testing.assert_allclose(loss, float(l2), rtol=1e-12)
    np.testing.assert_allclose(d, lt.grad.numpy(), atol=1e-12)


def test_adam_tf_first_step_known_answer():
    """beta1=0: first step = lr*sqrt(0.1)*g/(sqrt(0.1 g^2)+eps)  (SURVEY 8c)."""
    g = np.array([0.5, -2., 1e-3])
    p, m, v = R.adam_tf_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1, 2e-4)
    np.testing.assert_allclose(p, -2e-4 * np.sqrt(0.1) * g / (np.sqrt(0.1 * g * g) + 1e-8), rtol=1e-12)
    assert R.lr_decay(0) == 1. and R.lr_decay(25000) == 0.75 and R.lr_decay(50000) == 0.5


def test_param_counts_and_names():
    P = T.init_sngan_params(0)
    g = sum(v.size for k, v in P.items() if k.startswith('Generator/'))
    d = sum(v.size for k, v in P.items() if k.startswith('Discriminator/') and not T.is_state(k))
    assert g == 7875587 and d == 1701689
    assert sum(1 for k in P if T.is_state(k)) == 12
    assert 'Discriminator/D.Block.1.Conv1/filters/spectral_norm/u' in P
    assert 'Generator/G.Block.1.N1/CondBatchNorm/scale' in P
    assert 'Discriminator/Embedding.Label/embedding_map' in P


def test_network_shapes_and_finite_difference_of_d_loss():
    P = T.to_torch(T.init_sngan_params(1))
    rng = np.random.default_rng(1)
    z = t64(rng.normal(size=(4, 128)))
    labels = torch.tensor([1, 5, 0, 9])
    img = T.generator(P, z, labels, groups=2)
    assert img.shape == (4, 3072) and float(img.abs().max()) <= 1.
    real = torch.tensor(rng.integers(0, 256, (4, 3072)))
    deq = t64(rng.uniform(0, 1 / 128, (4, 3072)))
    loss, new_u, logits = T.d_loss_fn(P, real, labels, z, deq)
    assert logits.shape == (8,) and len(new_u) == 12
    name = 'Discriminator/D.Block.3.Conv1/Filters'
    (g,) = torch.autograd.grad(loss, P[name])
    idx = (1, 1, 3, 5)
    eps = 1e-5
    with torch.no_grad():
        P[name][idx] += eps
        lp = float(T.d_loss_fn(P, real, labels, z, deq)[0])
        P[name][idx] -= 2 * eps
        lm = float(T.d_loss_fn(P, real, labels, z, deq)[0])
        P[name][idx] += eps
    np.testing.assert_allclose((lp - lm) / (2 * eps), float(g[idx]), rtol=1e-4, atol=1e-9)


def test_batch_norm_train_known_answers():
    """normalization.py:8-24 restated: unit-variance output, and the zero-debiased moving mean of a constant batch
    mean IS that mean after any number of steps (that is what zero_debias_moving_mean buys)."""
    rng = np.random.default_rng(0)
    c = 5
    moving = dict(moving_mean=np.zeros(c), moving_variance=np.ones(c), biased=np.zeros(c), local_step=0.0)
    x = rng.normal(size=(6, 4, 4, c)) * 3.0 + 2.0
    for step in range(1, 4):
        y, _, moving = R.batch_norm_train(x, np.ones(c), np.zeros(c), moving)
        assert np.allclose(y.reshape(-1, c).mean(0), 0, atol=1e-12) and np.allclose(y.reshape(-1, c).var(0), 1, atol=1e-4)
        assert np.allclose(moving['moving_mean'], x.reshape(-1, c).mean(0), rtol=1e-12)
        vu = x.reshape(-1, c).var(0) * 96 / 95
        assert np.allclose(moving['moving_variance'], 1 + (1 - 0.9 ** step) * (vu - 1), rtol=1e-12)


def test_layer_instance_pixel_norm_gradients_match_autograd():
    """hand-derived backward passes of the three remaining normalisations (normalization.py:62-140) vs torch autograd"""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(3, 4, 5, 8)) * 1.7 + 0.3
    dy = rng.normal(size=x.shape)
    gamma, beta = rng.normal(size=8) + 1.0, rng.normal(size=8)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    # layer norm
    y, cache = R.layer_norm_forward(x, gamma, beta)
    mu = xt.mean(dim=(1, 2, 3), keepdim=True)
    var = ((xt - mu) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    yt = (xt - mu) / torch.sqrt(var + 1e-12) * gt + bt
    assert np.allclose(y, yt.detach().numpy(), atol=1e-12)
    gx, gg, gb = torch.autograd.grad(yt, [xt, gt, bt], torch.tensor(dy))
    dx, dg, db = R.layer_norm_backward(dy, gamma, cache)
    assert np.allclose(dx, gx.numpy(), atol=1e-10) and np.allclose(dg, gg.numpy(), atol=1e-10) and np.allclose(db, gb.numpy(), atol=1e-10)
    # instance norm: per (sample, channel) statistics over (H, W)
    y, _ = R.instance_norm_forward(x, gamma, beta)
    mu = xt.mean(dim=(1, 2), keepdim=True)
    var = ((xt - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    assert np.allclose(y, ((xt - mu) / torch.sqrt(var + 1e-6) * gt + bt).detach().numpy(), atol=1e-12)
    # pixel norm
    y = R.pixel_norm_forward(x)
    yt = xt * torch.rsqrt((xt * xt).mean(dim=3, keepdim=True) + 1e-8)
    assert np.allclose(y, yt.detach().numpy(), atol=1e-12)
    (gx,) = torch.autograd.grad(yt, [xt], torch.tensor(dy))
    assert np.allclose(R.pixel_norm_backward(dy, x), gx.numpy(), atol=1e-10)
    assert np.allclose((R.pixel_norm_forward(x) ** 2).mean(-1), 1.0, atol=1e-6)      # unit mean square per pixel


def test_bf16_storage_sensitivity_of_generator_gradients():
    """What bf16 STORAGE of activations and activation gradients alone does to the generator's gradients, with exact
    (float64) arithmetic everywhere else: the float64 oracle against itself with every tensor the product path
    materialises rounded to bf16 (ref_torch.STORE).  The deviation grows by 1-2 % (relative L2) per conditional batch
    norm on the way back from the image -- relu masks taken right after a normalisation flip for ~0.1 % of the
    elements, each flip costs the element's whole gradient -- and reaches ~0.1 at G.Input.  This is the floor for ANY
    bf16 implementation of this path (BASELINE.json config 2 is bf16); tests/test_model_gpu.py bounds the HIP path's
    per-tensor gradient errors at the headline batch against it."""
    import torch
    from oracle import ref_torch as T
    torch.set_num_threads(8)
    b = 8
    state = T.init_sngan_params(21)
    rng = np.random.default_rng(64)
    z2 = torch.tensor(rng.normal(size=(2 * b, 128))).to(torch.bfloat16).to(torch.float64)
    fl = torch.tensor(rng.integers(0, 10, 2 * b))

    def grads(store):
        T.STORE = store
        try:
            P = T.to_torch(state)
            loss, _ = T.g_loss_fn(P, z2, fl)
            gn = [k for k in T.trainable_names(P, 'Generator') if not (k.endswith('Biases') and 'G.Output' not in k)]
            return loss.item(), dict(zip(gn, torch.autograd.grad(loss, [P[k] for k in gn])))
        finally:
            T.STORE = None
    l0, g0 = grads(None)
    l1, g1 = grads(T.bf16_storage)
    assert abs(l0 - l1) < 5e-3
    err = {k: float((g1[k] - g0[k]).norm() / g0[k].norm()) for k in g0}
    assert err['Generator/G.Output/Filters'] < 0.02, err['Generator/G.Output/Filters']
    assert 0.02 < err['Generator/G.Input/W'] < 0.4, err['Generator/G.Input/W']            # measured 0.12 at batch 16, 0.125-0.15 at 64
    assert err['Generator/G.Input/W'] > 3 * err['Generator/G.Output/Filters']          # it accumulates with depth


def test_bf16_storage_sensitivity_of_acgan_gradients():
    """The same floor for BASELINE config 3 (ACGAN: batch norm + leaky relu critic, WGAN-GP through a double backward): the
    float64 restatement against itself with every stored tensor rounded to bf16.  Critic gradients move by up to ~0.12
    relative L2 (the 3x3 filters under a batch norm, two backward passes), generator gradients by ~0.2 all the way down (they
    cross the critic's 7 batch norms and their own 7).  tests/test_acgan_gpu.py bounds the HIP path per tensor against THIS
    floor computed on its own inputs, instead of an absolute number."""
    import torch
    from oracle import ref_torch as T
    torch.set_num_threads(8)
    batch = 8
    state = T.init_acgan_params(4)
    rng = np.random.default_rng(batch)
    real = torch.tensor(np.clip(rng.normal(size=(batch, 32, 32, 3)) * 0.5, -1, 1)).to(torch.bfloat16).double()
    rl = torch.tensor(rng.integers(0, 10, batch))
    z = torch.tensor(rng.normal(size=(batch, 128))).to(torch.bfloat16).double()
    fl = torch.tensor(rng.integers(0, 10, batch))
    alpha = torch.tensor(rng.uniform(size=batch))

    def grads(store):
        T.STORE = store
        try:
            P = T.to_torch(state)
            ld, _ = T.acgan_d_loss(P, real, rl, z, fl, alpha)
            dn = T.trainable_names(P, 'd_net')
            gd = dict(zip(dn, torch.autograd.grad(ld, [P[k] for k in dn])))
            P = T.to_torch(state)
            lg, _ = T.acgan_g_loss(P, z, fl)
            gn = T.trainable_names(P, 'g_net')
            return float(ld), gd, float(lg), dict(zip(gn, torch.autograd.grad(lg, [P[k] for k in gn])))
        finally:
            T.STORE = None
    ld0, d0, lg0, g0 = grads(None)
    ld1, d1, lg1, g1 = grads(T.bf16_storage)
    assert abs(ld0 - ld1) < 0.03 * max(1.0, abs(ld0)) and abs(lg0 - lg1) < 5e-3
    ed = {k: float((d1[k] - d0[k]).norm() / d0[k].norm()) for k in d0 if float(d0[k].norm()) > 1e-9}
    eg = {k: float((g1[k] - g0[k]).norm() / g0[k].norm()) for k in g0 if float(g0[k].norm()) > 1e-9}
    assert 0.03 < max(ed.values()) < 0.3, max(ed.values())                 # measured 0.12-0.2 at batch 8, 0.115 at 32
    assert 0.1 < eg['g_net/G.Input/W'] < 0.45, eg['g_net/G.Input/W']       # measured 0.23 at batch 8, 0.21 at 32
    assert eg['g_net/G.Output/Filters'] < eg['g_net/G.Input/W']            # it accumulates with depth


def test_acgan_restatement_known_answers():
    """ACGAN additions of the oracle (BASELINE config 3): train-mode batch norm equals torch's own functional form; the
    gradient-penalty expression has the closed form 10 (||w|| - 1)^2 for a linear critic; variable names and counts of
    ACGAN/model.py:21-90 (no spectral norm: no `u` vectors; batch-norm moving statistics are state, not parameters)."""
    import torch
    import torch.nn.functional as tF
    from oracle import ref_torch as T
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.normal(size=(5, 4, 4, 6)))
    g, b = torch.tensor(rng.normal(size=6)), torch.tensor(rng.normal(size=6))
    ref = tF.batch_norm(x.permute(0, 3, 1, 2), None, None, g, b, training=True, eps=1e-5).permute(0, 2, 3, 1)
    assert float((T.batch_norm_train(x, g, b) - ref).abs().max()) < 1e-12
    w = torch.tensor(rng.normal(size=(4 * 4 * 6,)))
    xi = x.clone().requires_grad_(True)
    (gr,) = torch.autograd.grad((xi.reshape(5, -1) @ w).sum(), xi, create_graph=True)
    slopes = torch.sqrt((gr ** 2).sum(dim=(1, 2, 3)) + 1e-10)
    assert abs(float(10. * ((slopes - 1.) ** 2).mean()) - 10. * (float(w.norm()) - 1.) ** 2) < 1e-9
    P = T.init_acgan_params(0)
    assert not any(k.endswith('spectral_norm/u') for k in P)
    gcount = sum(v.size for k, v in P.items() if k.startswith('g_net/') and not T.is_state(k))
    dcount = sum(v.size for k, v in P.items() if k.startswith('d_net/') and not T.is_state(k))
    # generator: SNGAN's 7 875 587 with G.OutputNorm's 10 x 256 x 2 table replaced by one gamma/beta pair
    assert gcount == 7875587 - 2 * 10 * 256 + 2 * 256
    conv = lambda k, ci, co: k * k * ci * co + co      # noqa: E731
    assert dcount == conv(1, 3, 128) + conv(3, 3, 128) + conv(3, 128, 128) + conv(1, 128, 128) + 6 * conv(3, 128, 128) + 6 * 2 * 128 + 129 + 1290
    assert T.is_state('d_net/D.NoneBlock.3.N1/BatchNorm/moving_mean/local_step') and not T.is_state('d_net/D.NoneBlock.3.N1/BatchNorm/gamma')


def test_pggan_oracle_pieces():
    """oracle/ref_pggan.py: the minibatch-std statistic against plain NumPy, the TF-1.5 bilinear resize on known values, and
    shapes / variable names of the restated networks at three stages of the progression."""
    from oracle import ref_pggan as G
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 4, 4, 7))
    assert np.allclose(G.minibatch_std(torch.tensor(x)).numpy(), G.minibatch_std_numpy(x))
    assert np.allclose(G.minibatch_std_numpy(np.ones((3, 2, 2, 4)))[..., -1], 1e-4)            # zero variance: sqrt(1e-8)
    r = G.resize_bilinear(np.arange(16.).reshape(1, 4, 4, 1), (8, 8))[0, :, :, 0]
    assert r[0, 1] == 0.5 and r[1, 0] == 2.0 and r[2, 2] == 5.0 and r[7, 7] == 15.0
    assert np.array_equal(G.resize_bilinear(np.arange(64.).reshape(1, 8, 8, 1), (4, 4))[0, :, :, 0], np.arange(64.).reshape(8, 8)[::2, ::2])
    y = torch.tensor(rng.normal(size=(2, 3, 3, 6)))
    assert torch.allclose((G.pixel_norm(y) ** 2).mean(dim=3), torch.ones(2, 3, 3, dtype=torch.float64), atol=1e-6)
    assert [G.get_dim(i) for i in range(7)] == [512, 512, 512, 256, 128, 64, 32]
    for bc, trans in ((0, False), (1, True), (2, False)):
        P = T.to_torch(G.init_params(1, bc, trans, z_dim=32))
        z = torch.tensor(rng.normal(size=(3, 32)))
        img = G.generator(P, z, 0.25, bc, trans)
        assert img.shape == (3, 4 * 2 ** bc, 4 * 2 ** bc, 3)
        lg, nu = G.discriminator(P, img, 0.25, bc, trans, update_u=True)
        assert lg.shape == (3,) and len(nu) == sum(1 for k in P if k.endswith('spectral_norm/u'))
        if trans:      # alpha = 0: only the skip path (toRGB2 of the upsampled features); alpha = 1: only the new block
            a0, a1 = G.generator(P, z, 0.0, bc, trans), G.generator(P, z, 1.0, bc, trans)
            assert torch.allclose(img, 0.75 * a0 + 0.25 * a1, atol=1e-9)


def test_pix2pix_oracle_pieces():
    """oracle/ref_pix2pix.py: tf.nn.conv2d semantics for even filters (SAME puts the surplus pad behind, tf.pad + VALID, the
    1x1 -> 1x1 stride-2 bottom) against plain loops, instance norm moments, and the variable list of the U-Net / PatchGAN."""
    from oracle import ref_pix2pix as X
    rng = np.random.default_rng(0)
    w = rng.normal(size=(4, 4, 3, 2))
    for h, stride, padding, pad_in, pad, out in ((5, 2, 'SAME', 0, 1, 3), (6, 2, 'SAME', 0, 1, 3), (1, 2, 'SAME', 0, 1, 1), (5, 1, 'SAME', 0, 1, 5),
                                                 (5, 1, 'VALID', 1, 1, 4), (6, 2, 'VALID', 1, 1, 3)):
        x = rng.normal(size=(2, h, h, 3))
        y = X.conv2d_tf(torch.tensor(x), torch.tensor(w), None, stride, padding, pad_in).numpy()
        assert y.shape == (2, out, out, 2), (h, stride, padding, y.shape)
        assert np.allclose(y, X.conv2d_numpy(x, w, None, stride, pad, (out, out)))
    x = torch.tensor(rng.normal(size=(3, 4, 4, 5)) * 2 + 1)
    y = X.instance_norm(x, torch.ones(1, 5, dtype=torch.float64), torch.zeros(1, 5, dtype=torch.float64))
    assert torch.allclose(y.mean(dim=(1, 2)), torch.zeros(3, 5, dtype=torch.float64), atol=1e-9)
    assert torch.allclose((y ** 2).mean(dim=(1, 2)), torch.ones(3, 5, dtype=torch.float64), atol=1e-4)
    one = X.instance_norm(x[:, :1, :1], torch.ones(1, 5, dtype=torch.float64), torch.full((1, 5), 0.3, dtype=torch.float64))
    assert torch.allclose(one, torch.full_like(one, 0.3))                    # a 1x1 map normalises to its offset (the U-Net's bottom)
    P = X.init_params(0, ngf=8, ndf=8)
    assert sum(1 for k in P if k.endswith('/Filters')) == 9 + 9 + 6 and sum(1 for k in P if k.endswith('spectral_norm/u')) == 6
    assert P['g_net/decoder_8/Conv2D/Filters'].shape == (4, 4, 128, 64) and P['g_net/decoder_1/Conv2D/Filters'].shape == (4, 4, 16, 3)
    assert P['d_net/layer_1/Conv2D/Filters'].shape == (4, 4, 6, 8) and P['d_net/layer_6/Conv2D/Filters'].shape == (4, 4, 64, 1)
