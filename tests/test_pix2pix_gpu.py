"""Pix2Pix configuration (BASELINE.json config 5; Pix2Pix/networks.py U-Net + PatchGAN, Pix2Pix/train.py create_model) on a
real MI355X against the float64 oracle (oracle/ref_pix2pix.py): the general convolution (4x4 filters, stride 1 / 2, TF SAME
and tf.pad + VALID, fused relu / NN-upsampling / tanh) forward and all three gradients, channel concat, dropout, L1 loss, the
two networks, both losses with every gradient and the spectral-norm `u` policy, and training steps.  The reference U-Net has
nine stride-2 encoders, so its skip connections only line up for inputs that are multiples of 512 (its default crop_size):
network-level cases run at 512 x 512.  bf16 activations / fp32 accumulate; tolerances stated at each assertion."""
import numpy as np
import pytest
import torch

from oracle import ref_pix2pix as X
from oracle import ref_torch as T

pytestmark = pytest.mark.gpu
BF_TOL, F32_FROM_BF_TOL = 1e-2, 2e-3


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
    ref = torch.as_tensor(ref, dtype=torch.float64).detach()
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-300))


def l2(got, ref):
    got, ref = got.detach().to(torch.float64).cpu().flatten(), ref.detach().to(torch.float64).flatten()
    assert torch.isfinite(got).all()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-300))


def cos(got, ref):
    got, ref = got.detach().to(torch.float64).cpu().flatten(), ref.detach().to(torch.float64).flatten()
    return float((got @ ref) / max(float(got.norm() * ref.norm()), 1e-300))


@pytest.mark.parametrize("n,h,cin,cout,stride,padding,pad_input,up,relu,tanh", [
    (2, 16, 3, 64, 2, 'SAME', 0, False, False, False),        # encoder_1: narrow input
    (2, 16, 64, 128, 2, 'SAME', 0, False, False, False),      # encoder
    (3, 1, 128, 128, 2, 'SAME', 0, False, False, False),      # the 1x1 -> 1x1 bottom of the U-Net: only the centre taps see data
    (2, 2, 128, 64, 2, 'SAME', 0, False, False, False),
    (2, 8, 128, 64, 1, 'SAME', 0, True, True, False),         # decoder: relu + NN-upsample + 4x4 SAME (pad 1 / 2)
    (2, 16, 128, 3, 1, 'SAME', 0, True, True, True),          # decoder_1: 3 output channels, tanh
    (2, 32, 6, 64, 2, 'VALID', 1, False, False, False),       # critic layer_1: tf.pad 1 + VALID
    (2, 16, 64, 128, 1, 'VALID', 1, False, False, False),     # critic layer_5: 16 -> 15
    (2, 15, 128, 1, 1, 'VALID', 1, False, False, False),      # critic layer_6: 15 -> 14, one output channel
    (4, 64, 64, 32, 1, 'SAME', 0, True, True, False),         # a decoder large enough for the phase-stacked form (one 3x3 conv to 4 Cout + depth_to_space)
    (1, 256, 3, 64, 2, 'SAME', 0, False, False, False),       # encoder_1 at a size whose filter gradient takes the im2col + 1x1 route
    (1, 256, 6, 64, 2, 'VALID', 1, False, False, False),      # critic layer_1 likewise (96 im2col columns)
])
def test_general_conv_forward_and_gradients(gpu, n, h, cin, cout, stride, padding, pad_input, up, relu, tanh):
    from gan_lib_tensorflow_amd.common.ops import conv2d as C
    from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
    rng = np.random.default_rng(h * 7 + cin + cout + stride)
    store = set_default_store(ParamStore("cuda", seed=1))
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    xt.requires_grad_(True)
    y = C.Conv2D(xt, cin, cout, 4, stride, 'L', padding=padding, he_init=True, biases=True, upsample=up, in_relu=relu, out_tanh=tanh, pad_input=pad_input)
    W, b = store.vars['L/Filters'], store.vars['L/Biases']
    with torch.no_grad():
        b.copy_(torch.tensor(rng.normal(size=cout), dtype=torch.float32))
        W.copy_(W.to(torch.bfloat16).float())                       # bf16-representable weights: the comparison isolates the arithmetic
    y = C.Conv2D(xt, cin, cout, 4, stride, 'L', padding=padding, he_init=True, biases=True, upsample=up, in_relu=relu, out_tanh=tanh, pad_input=pad_input)
    xr = x.clone().requires_grad_(True)
    wr, br = W.detach().double().cpu().requires_grad_(True), b.detach().double().cpu().requires_grad_(True)
    hin = torch.relu(xr) if relu else xr
    hin = T.upsample_nn2x(hin) if up else hin
    ref = X.conv2d_tf(hin, wr, br, stride, padding, pad_input)
    ref = torch.tanh(ref) if tanh else ref
    assert tuple(y.shape) == tuple(ref.shape), (y.shape, ref.shape)
    assert rel(y, ref) < BF_TOL
    if n * h * h * cin <= 4096:        # plain-loop restatement of the gather on the small cases
        xin = hin.detach().numpy()
        pad = pad_input if padding == 'VALID' else max((ref.shape[1] - 1) * stride + 4 - xin.shape[1], 0) // 2
        chk = X.conv2d_numpy(xin, wr.detach().numpy(), br.detach().numpy(), stride, pad, ref.shape[1:3])
        assert np.allclose(np.tanh(chk) if tanh else chk, ref.detach().numpy(), atol=1e-9)
    dy, dyt = bf(rng.normal(size=ref.shape))
    ref.backward(dy)
    y.backward(dyt)
    torch.cuda.synchronize()
    assert l2(xt.grad, xr.grad) < BF_TOL
    assert l2(W.grad, wr.grad) < F32_FROM_BF_TOL * 2 and l2(b.grad, br.grad) < F32_FROM_BF_TOL * 2


def test_concat_dropout_l1(gpu):
    from gan_lib_tensorflow_amd import functional as Fn, kernels as K
    rng = np.random.default_rng(1)
    a, at = bf(rng.normal(size=(3, 8, 8, 64)))
    b, bt = bf(rng.normal(size=(3, 8, 8, 24)))
    at.requires_grad_(True); bt.requires_grad_(True)
    y = Fn.concat_channels(at, bt)
    assert torch.equal(y.cpu(), torch.cat([at.detach().cpu(), bt.detach().cpu()], 3))
    g, gt = bf(rng.normal(size=y.shape))
    y.backward(gt)
    assert torch.equal(at.grad.cpu(), gt[..., :64].cpu()) and torch.equal(bt.grad.cpu(), gt[..., 64:].cpu())
    # odd widths take the element-wise form (PGGAN: 513 + 63 channels of zero padding)
    c, ct = bf(rng.normal(size=(2, 4, 4, 13)))
    d, dt = bf(rng.normal(size=(2, 4, 4, 3)))
    yo = K.concat_channels(ct, dt)
    assert torch.equal(yo.cpu(), torch.cat([ct.cpu(), dt.cpu()], 3))
    ca, cb = K.split_channels(yo, 13)
    assert torch.equal(ca, ct) and torch.equal(cb, dt)
    # dropout: Bernoulli(keep) mask from the device RNG, survivors scaled by 1/keep, a fresh mask per call, same mask backward
    rs = K.new_rng_state(11, "cuda")
    x, xt = bf(rng.normal(size=(4, 16, 16, 512)))
    xt.requires_grad_(True)
    y1 = Fn.dropout(xt, 0.5, rs)
    y2 = Fn.dropout(xt, 0.5, rs)
    kept = (y1 != 0).float().mean().item()
    assert abs(kept - 0.5) < 0.01 and not torch.equal(y1, y2)
    m = (y1 != 0)
    assert rel(y1[m], (2.0 * x.cuda()[m]).cpu()) < 1e-2
    y1.backward(torch.ones_like(y1))
    assert torch.equal(xt.grad != 0, m) and rel(xt.grad[m], torch.full((int(m.sum()),), 2.0)) < 1e-6     # rel() moves `got` to the host
    y3 = Fn.dropout(xt, 0.8, rs)
    assert abs((y3 != 0).float().mean().item() - 0.8) < 0.01
    # L1 loss (train.py:510) and its gradient inside a weighted sum
    p, pt = bf(rng.normal(size=(2, 16, 16, 3)))
    q, qt = bf(rng.normal(size=(2, 16, 16, 3)))
    pt.requires_grad_(True)
    loss = Fn.l1_loss(pt, qt)
    assert abs(float(loss) - float((p - q).abs().mean())) < 1e-5
    (loss * 100.0).backward()
    assert rel(pt.grad, 100.0 * torch.sign(p - q) / p.numel()) < 1e-2


class _MaskLog:
    """records the masks K.dropout_fwd draws, in call order (the oracle needs the same masks)"""

    def __init__(self, K):
        self.K, self.orig, self.masks = K, K.dropout_fwd, []

    def __enter__(self):
        def wrapped(x, keep, rng_state):
            y, mask = self.orig(x, keep, rng_state)
            self.masks.append(mask)
            return y, mask
        self.K.dropout_fwd = wrapped
        return self

    def __exit__(self, *exc):
        self.K.dropout_fwd = self.orig
        return False

    def as_oracle(self, start=0):
        names = ['g_net/decoder_9', 'g_net/decoder_8', 'g_net/decoder_7']
        return {nm: m.double().cpu() for nm, m in zip(names, self.masks[start:start + 3])}


def make(batch=1, size=512, seed=3):
    from gan_lib_tensorflow_amd.Pix2Pix.train import Pix2PixTrainer, default_args
    tr = Pix2PixTrainer(default_args(batch_size=batch, crop_size=size, max_steps=1000), seed=seed)
    return tr, tr.store.state_dict()


def _bad(tr, names, gref, cos_min, l2_max):
    bad = []
    gmax = max(float(g.abs().max()) for g in gref.values())
    for k in names:
        g, r = tr.store.vars[k].main_grad, gref[k]
        if float(r.abs().max()) < 1e-6 * gmax:          # e.g. a conv bias in front of an instance norm: exactly zero
            if float(g.abs().max()) > 2e-3 * gmax:
                bad.append((k, 'abs', float(g.abs().max())))
            continue
        c, e = cos(g, r), l2(g, r)
        if c < cos_min or e > l2_max:
            bad.append((k, round(c, 4), round(e, 3)))
    return bad


def test_pix2pix_networks_losses_gradients_vs_oracle(gpu):
    """U-Net output, PatchGAN logits (30 x 30 patches at 512 x 512), both losses, every gradient, and the spectral-norm `u`
    after the two critic passes of an update, against the float64 restatement from the same parameters, inputs and dropout
    masks.  Bounds: images |delta| <= 0.1 (tanh range) and mean <= 2x the bf16-storage floor of the restatement itself, patch logits <= 3e-2 * max|ref|; critic gradients cosine >= 0.99 /
    relative L2 <= 0.1; generator gradients against the bf16-storage floor of the restatement itself (see below; measured values printed)."""
    from gan_lib_tensorflow_amd import kernels as K
    tr, state = make()
    names = sorted(state)
    assert names == sorted(X.init_params(0)), set(names) ^ set(X.init_params(0))        # the reference's variable names
    P = T.to_torch(state)
    rng = np.random.default_rng(5)
    a, at = bf(np.clip(rng.normal(size=(1, 512, 512, 3)) * 0.5, -1, 1))
    b, bt = bf(np.clip(rng.normal(size=(1, 512, 512, 3)) * 0.5, -1, 1))
    with _MaskLog(K) as log, torch.no_grad():
        out = tr._generator(at)
        masks = log.as_oracle()
        out_ref = X.generator(P, a, masks)
        T.STORE = T.bf16_storage                   # the same graph with every stored tensor rounded to bf16: the storage floor
        try:
            out_bf = X.generator(P, a, masks)
        finally:
            T.STORE = None
    assert out.shape == (1, 512, 512, 3)
    d, floor = (out.double().cpu() - out_ref).abs(), (out_bf - out_ref).abs()
    print("pix2pix image max |delta|", float(d.max()), "mean", float(d.mean()), "| bf16-storage floor of the restatement: max", float(floor.max()),
          "mean", float(floor.mean()))
    # 17 convolutions and 16 instance norms (some over 2x2 and 4x4 maps, batch 1) between input and output: the error is that of
    # bf16 storage -- bounded by the floor the float64 graph shows when ITS stored tensors are rounded to bf16
    assert float(d.max()) < 0.1 and float(d.mean()) < max(2.0 * float(floor.mean()), 3e-3)
    with torch.no_grad():
        pr = tr._critic(at, bt, 'NO_OPS')
        pr_ref, _ = X.discriminator(P, a, b)
    assert pr.shape == (1, 30, 30, 1) and rel(pr, pr_ref) < 3e-2
    # critic loss
    with _MaskLog(K) as log:
        loss = tr.d_loss(at, bt)
        tr._backward(loss)
        masks = log.as_oracle()
    loss_ref, u_ref = X.d_loss(P, a, b, masks)
    dn = T.trainable_names(P, 'd_net')
    gref = dict(zip(dn, torch.autograd.grad(loss_ref, [P[k] for k in dn])))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-2 * max(1.0, abs(float(loss_ref)))
    assert _bad(tr, dn, gref, 0.99, 0.1) == []
    for k, u in u_ref.items():
        assert rel(tr.store.vars[k], u) < 2e-3, k                        # advanced twice: real pass, then fake pass
    tr.d_flat['grads'].zero_()
    # generator loss
    P = T.to_torch(tr.store.state_dict())
    with _MaskLog(K) as log:
        loss = tr.g_loss(at, bt)
        tr._backward(loss)
        masks = log.as_oracle()
    loss_ref, parts = X.g_loss(P, a, b, masks)
    gn = T.trainable_names(P, 'g_net')
    gref = dict(zip(gn, torch.autograd.grad(loss_ref, [P[k] for k in gn])))
    torch.cuda.synchronize()
    print("pix2pix g_loss", float(loss), float(loss_ref), "L1", float(tr.losses['gen_loss_L1']), float(parts['l1']))
    assert abs(float(loss) - float(loss_ref)) < 2e-2 * max(1.0, abs(float(loss_ref)))
    assert abs(float(tr.losses['gen_loss_L1']) - float(parts['l1'])) < 5e-3
    # The generator's gradient crosses the critic, 17 convolutions and 16 instance-norm backward passes (mean subtraction over
    # maps as small as 2x2, batch 1) with every tensor stored in bf16: the error grows layer by layer from 0.4 % at decoder_1
    # to ~30 % at the encoders.  The yardstick is the restatement itself with ITS stored tensors rounded to bf16 (forward and
    # backward): per tensor, the HIP path may be at most 1.5x as far from the float64 gradient as that floor (+ 0.05).
    T.STORE = T.bf16_storage
    try:
        P2 = T.to_torch(tr.store.state_dict())
        # (u was advanced by tr.g_loss: rewind the restatement's copy to what the float64 run read)
        for k in P2:
            if k.endswith('spectral_norm/u'):
                P2[k] = P[k].clone()
        loss_bf, _ = X.g_loss(P2, a, b, masks)
        gfloor = dict(zip(gn, torch.autograd.grad(loss_bf, [P2[k] for k in gn])))
    finally:
        T.STORE = None
    gmax = max(float(g.abs().max()) for g in gref.values())
    bad, table = [], {}
    for k in gn:
        g, r = tr.store.vars[k].main_grad, gref[k]
        if float(r.abs().max()) < 1e-6 * gmax:          # a conv bias in front of an instance norm: exactly zero
            if float(g.abs().max()) > 2e-3 * gmax:
                bad.append((k, 'abs', float(g.abs().max())))
            continue
        c, e, f = cos(g, r), l2(g, r), l2(gfloor[k], r)
        table[k.split('/', 1)[1]] = (round(c, 4), round(e, 3), round(f, 3))
        if c < 0.9 or e > 1.5 * f + 0.05:
            bad.append((k, c, e, f))
    print("pix2pix G grads (cosine, relative L2, bf16-storage floor of the restatement):", table)
    assert not bad, bad
    assert all(float(tr.store.vars[k].main_grad.abs().max()) == 0.0 for k in dn)      # gen_loss moves g_vars only (train.py:552)


def test_pix2pix_training_steps(gpu, deterministic_stats):
    """train.py:704-730: n_dis critic updates then one generator update per step; learning rate 2e-4 -> 1e-4 over max_steps;
    parameters move by at most ~lr per update and stay finite."""
    from gan_lib_tensorflow_amd.Pix2Pix.train import polynomial_decay
    assert polynomial_decay(0, 2e-4, 1000, 1e-4) == 2e-4 and abs(polynomial_decay(500, 2e-4, 1000, 1e-4) - 1.5e-4) < 1e-12
    tr, _ = make(batch=2, seed=7)
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(2, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    b = (torch.rand(2, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    p0d, p0g = tr.d_flat['params'].clone(), tr.g_flat['params'].clone()
    for _ in range(2):
        tr.train_step(a, b)
    torch.cuda.synchronize()
    assert tr.global_step == 2 and int(tr.d_opt['t']) == 10 and int(tr.g_opt['t']) == 2
    for flat, p0, n in ((tr.d_flat, p0d, 10), (tr.g_flat, p0g, 2)):
        assert bool(torch.isfinite(flat['params']).all()) and float(flat['grads'].abs().max()) == 0.0
        moved = (flat['params'] - p0).abs()
        assert 1e-5 < float(moved.max()) < n * 2e-4 * 1.5
    assert all(np.isfinite(float(v)) for v in tr.losses.values())


def test_pix2pix_batch16_critic_pass_is_the_mean_of_its_sub_batches(gpu):
    """BASELINE.json config 5 at its full size (batch 16, 512 x 512): every layer of the PatchGAN critic is per sample (instance
    norm), so the logits of a batch are the logits of its samples and the gradient of a mean loss over 16 pairs is the mean of the
    gradients over four sub-batches of 4 -- a size-independent check of the general conv kernels' forward pass and all three
    gradients at shapes the float64 restatement cannot reach in a test."""
    from gan_lib_tensorflow_amd import functional as Fn
    tr, _ = make(batch=16, seed=11)
    g = torch.Generator().manual_seed(8)
    a = (torch.rand(16, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    b = (torch.rand(16, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    logits = tr._critic(a, b, 'NO_OPS')
    assert logits.shape == (16, 30, 30, 1)
    tr._backward(Fn.hinge_g_loss(logits.reshape(-1)))
    torch.cuda.synchronize()
    g16 = tr.d_flat['grads'].clone()
    assert bool(torch.isfinite(g16).all()) and float(g16.abs().max()) > 0
    tr.d_flat['grads'].zero_()
    parts = []
    for i in range(4):
        lg = tr._critic(a[4 * i:4 * i + 4].contiguous(), b[4 * i:4 * i + 4].contiguous(), 'NO_OPS')
        parts.append(lg.detach())
        tr._backward(Fn.hinge_g_loss(lg.reshape(-1)))
    torch.cuda.synchronize()
    assert rel(logits, torch.cat(parts, 0).double().cpu()) < 1e-2
    g4 = tr.d_flat['grads'] / 4
    assert l2(g16, g4.double().cpu()) < 1e-2, l2(g16, g4.double().cpu())
    tr.d_flat['grads'].zero_()


def test_pix2pix_batch16_generator_pass_is_the_mean_of_its_sub_batches(gpu):
    """BASELINE.json config 5 at its full size on the OTHER network: the U-Net generator at batch 16, 512 x 512.  Every layer is
    per sample (instance norm; the decoder's dropout masks are recorded from the batch-16 pass and re-applied slice by slice),
    so the images of the batch are the images of its four sub-batches of 4 and the gradient of a mean L1 loss over 16 pairs is
    the mean of the four sub-batch gradients.  At batch 16 the dispatcher takes routes batch 1-4 never take (decoder_3 / decoder_4
    cross the phase-stack threshold of functional.PHASE_STACK_MIN_PIXELS, other tile shapes): a size-independent check of those
    routes' forward pass and all three gradients, at a size the float64 restatement cannot reach in a test."""
    from gan_lib_tensorflow_amd import functional as Fn, kernels as K
    tr, _ = make(batch=16, seed=13)
    g = torch.Generator().manual_seed(9)
    a = (torch.rand(16, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    b = (torch.rand(16, 512, 512, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    with _MaskLog(K) as log:
        out16 = tr._generator(a)
        tr._backward(Fn.l1_loss(out16, b))
    torch.cuda.synchronize()
    assert out16.shape == (16, 512, 512, 3) and len(log.masks) == 3
    g16 = tr.g_flat['grads'].clone()
    assert bool(torch.isfinite(g16).all()) and float(g16.abs().max()) > 0
    tr.g_flat['grads'].zero_()
    masks, orig, parts = log.masks, K.dropout_fwd, []
    try:
        for i in range(4):
            calls = iter(range(3))

            def replay(x, keep, rng_state, i=i, calls=calls):         # the recorded mask rows of this sub-batch, same kernel arithmetic
                m = masks[next(calls)][4 * i:4 * i + 4].contiguous()
                return K.dropout_bwd(x, m, keep), m
            K.dropout_fwd = replay
            o = tr._generator(a[4 * i:4 * i + 4].contiguous())
            parts.append(o.detach())
            tr._backward(Fn.l1_loss(o, b[4 * i:4 * i + 4].contiguous()))
    finally:
        K.dropout_fwd = orig
    torch.cuda.synchronize()
    # images in tanh range: the two batch sizes run different kernels (other tiles, the phase-stacked decoders), bf16 rounding apart
    d = (out16.double().cpu() - torch.cat(parts, 0).double().cpu()).abs()
    assert float(d.max()) < 0.1 and float(d.mean()) < 4e-3, (float(d.max()), float(d.mean()))
    g4 = tr.g_flat['grads'] / 4
    e = l2(g16, g4.double().cpu())
    assert e < 3e-2, e
    tr.g_flat['grads'].zero_()


def test_weight_side_transform_kernels_vs_torch(gpu):
    """gank_phase_stack4 (+ adjoint), gank_pad_rows (+ adjoint, fp32 and 16-bit), gank_tile_rows (+ adjoint), gank_fewout_pack
    (+ adjoint), gank_zero_f32: the library kernels that replaced torch.einsum / pad / repeat / add_ on the Pix2Pix / PGGAN routes,
    against those torch expressions (the checker here; sums of <= 4 fp32 values, exact or to 1 ulp)."""
    from gan_lib_tensorflow_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.tensor([[[1., 0, 0, 0], [0, 1, 1, 0], [0, 0, 0, 1]], [[0., 0, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1]]], device="cuda")      # [a][u][ky]
    for cin, cout in ((64, 32), (128, 96)):
        w4 = torch.randn((4, 4, cin, cout), generator=g, device="cuda")
        w3 = K.phase_stack4(w4)
        ref = torch.einsum('auk,bvl,klio->uviabo', A, A, w4).reshape(3, 3, cin, 4 * cout)
        assert float((w3 - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        g3 = torch.randn((3, 3, cin, 4 * cout), generator=g, device="cuda")
        base = torch.randn((4, 4, cin, cout), generator=g, device="cuda")
        dw = K.phase_stack4_bwd(g3, base.clone())
        refb = base + torch.einsum('auk,bvl,uviabo->klio', A, A, g3.reshape(3, 3, cin, 2, 2, cout))
        assert float((dw - refb).abs().max()) <= 2e-6 * float(refb.abs().max())
        # adjointness: <stack(w), g> == <w, fold(g)>
        lhs, rhs = float((w3.double() * g3.double()).sum()), float((w4.double() * (dw - base).double()).sum())
        assert abs(lhs - rhs) < 1e-4 * (abs(lhs) + 1)
    w = torch.randn((3, 3, 513, 64), generator=g, device="cuda")
    wp = K.pad_rows(w.view(9, 513 * 64), 576 * 64).view(3, 3, 576, 64)
    assert torch.equal(wp, torch.nn.functional.pad(w, (0, 0, 0, 63)))
    gw = torch.randn((3, 3, 576, 64), generator=g, device="cuda")
    acc = torch.ones((3, 3, 513, 64), device="cuda")
    K.pad_rows_bwd(gw.view(9, 576 * 64), acc.view(9, 513 * 64))
    assert torch.equal(acc, 1 + gw[:, :, :513, :])
    x = torch.randn((3, 5, 7, 513), generator=g, device="cuda").to(torch.bfloat16)
    xp = K.pad_rows(x, 576)
    assert torch.equal(xp[..., :513], x) and float(xp[..., 513:].float().abs().max()) == 0.0
    b = torch.randn(96, generator=g, device="cuda")
    assert torch.equal(K.tile_rows(b, 4), b.repeat(4))
    gb = torch.randn(4 * 96, generator=g, device="cuda")
    db = torch.ones(96, device="cuda")
    K.tile_rows_bwd(gb, db, 4)
    assert float((db - (1 + gb.view(4, 96).sum(0))).abs().max()) < 1e-5
    wf = torch.randn((4, 4, 128, 3), generator=g, device="cuda")
    wz = K.fewout_pack(wf, 64)
    assert torch.equal(wz, torch.nn.functional.pad(wf.permute(2, 0, 1, 3).reshape(128, 48), (0, 16)))
    gz = torch.randn((128, 64), generator=g, device="cuda")
    dwf = torch.ones((4, 4, 128, 3), device="cuda")
    K.fewout_pack_bwd(gz, dwf)
    assert torch.equal(dwf, 1 + gz[:, :48].reshape(128, 16, 3).permute(1, 0, 2).reshape(4, 4, 128, 3))
    z = K.zeros_f32((3, 1000), "cuda")
    assert float(z.abs().max()) == 0.0 and z.shape == (3, 1000)


def test_one_element_per_thread_kernels_cover_tensors_beyond_the_grid_cap(gpu):
    """gank_depth_to_space2 / _space_to_depth2, gank_im2col_narrow and the tap gather / scatter pair take one 16-byte piece (or pixel)
    per thread; launched on the capped grid of the grid-stride kernels (4096 blocks) they left everything behind the first 8.4 M
    elements unwritten -- every Pix2Pix pass at batch >= 4 went through such tensors.  Sizes here are 2.5x past that cap; the
    outputs are compared in full (torch on the GPU as the checker: these are pure data movements / short sums)."""
    from gan_lib_tensorflow_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((5, 128, 128, 256), generator=g, device="cuda").to(torch.bfloat16)           # 21 M elements
    y = K.depth_to_space2(x)
    ref = x.view(5, 128, 128, 2, 2, 64).permute(0, 1, 3, 2, 4, 5).reshape(5, 256, 256, 64)
    assert torch.equal(y, ref) and torch.equal(K.space_to_depth2(y), x)
    img = torch.randn((5, 512, 512, 6), generator=g, device="cuda").to(torch.bfloat16)
    col = K.im2col_narrow(img, (256, 256), 4, 2, 1, 128)                                           # 42 M elements
    pad = torch.nn.functional.pad(img, (0, 0, 1, 1, 1, 1))
    for (ky, kx) in ((0, 0), (1, 2), (3, 3)):
        want = pad[:, ky:ky + 512:2, kx:kx + 512:2, :]
        assert torch.equal(col[..., (ky * 4 + kx) * 6:(ky * 4 + kx) * 6 + 6], want), (ky, kx)
    assert float(col[..., 96:].abs().max()) == 0.0
    z = torch.randn((5, 256, 256, 64), generator=g, device="cuda").to(torch.bfloat16)            # 21 M elements in, 3.9 M pixels out
    out = K.tap_gather_up2(z, None, 4, 1, 3, False)
    assert out.shape == (5, 512, 512, 3) and bool(torch.isfinite(out.float()).all())
    gsm = torch.randn((5, 512, 512, 3), generator=g, device="cuda").to(torch.bfloat16)
    colg = K.tap_scatter_up2(gsm, 4, 1, 64)
    # adjointness over the WHOLE tensors: <scatter(g), z> == <g, gather(z)> (an unwritten tail breaks it)
    lhs, rhs = float((colg.double() * z.double()).sum()), float((gsm.double() * out.double()).sum())
    assert abs(lhs - rhs) < 2e-2 * float((colg.double() * z.double()).abs().sum()) and abs(lhs) > 0
    # and the tails themselves: the last sample equals the same sample run alone
    assert torch.equal(K.tap_gather_up2(z[4:].contiguous(), None, 4, 1, 3, False), out[4:])
    assert torch.equal(K.tap_scatter_up2(gsm[4:].contiguous(), 4, 1, 64), colg[4:])


def test_im2col_depth_to_space_and_tap_kernels_against_numpy(gpu):
    """The data-movement kernels behind the round-3 Pix2Pix routes, each against a NumPy restatement: gank_im2col_narrow (bit for bit:
    a gather of bf16 values), gank_depth_to_space2 / gank_space_to_depth2 (bit for bit, and inverse of each other), gank_tap_scatter_up2
    (sums of at most 4 bf16 values in fp32, rounded once) and gank_tap_gather_up2 (sums of k*k partials + bias, tanh)."""
    from gan_lib_tensorflow_amd import kernels as K
    rng = np.random.default_rng(12)
    # im2col: 4x4 stride 2 pad 1 on a 6-channel image (critic layer_1), 96 columns padded to 128
    x, xt = bf(rng.normal(size=(2, 10, 12, 6)))
    col = K.im2col_narrow(xt, (5, 6), 4, 2, 1, 128)
    ref = np.zeros((2, 5, 6, 128))
    xn = x.numpy()
    for oy in range(5):
        for ox in range(6):
            for ky in range(4):
                for kx in range(4):
                    iy, ix = 2 * oy - 1 + ky, 2 * ox - 1 + kx
                    if 0 <= iy < 10 and 0 <= ix < 12:
                        ref[:, oy, ox, (ky * 4 + kx) * 6:(ky * 4 + kx) * 6 + 6] = xn[:, iy, ix, :]
    torch.cuda.synchronize()
    assert np.array_equal(col.double().cpu().numpy(), ref)
    # depth_to_space / space_to_depth
    y3, y3t = bf(rng.normal(size=(2, 3, 5, 4 * 16)))
    y = K.depth_to_space2(y3t)
    refy = y3.numpy().reshape(2, 3, 5, 2, 2, 16).transpose(0, 1, 3, 2, 4, 5).reshape(2, 6, 10, 16)
    torch.cuda.synchronize()
    assert np.array_equal(y.double().cpu().numpy(), refy) and torch.equal(K.space_to_depth2(y), y3t)
    # tap scatter / gather for a 4x4 window with 1 leading pad behind a 2x upsample, 3 output channels, 64 columns
    g, gt = bf(rng.normal(size=(2, 8, 12, 3)))
    colg = K.tap_scatter_up2(gt, 4, 1, 64)
    refc = np.zeros((2, 4, 6, 64))
    gn = g.numpy()
    for qy in range(4):
        for qx in range(6):
            for ky in range(4):
                for kx in range(4):
                    for a in range(2):
                        for b_ in range(2):
                            py, px = 2 * qy + a - (ky - 1), 2 * qx + b_ - (kx - 1)
                            if 0 <= py < 8 and 0 <= px < 12:
                                refc[:, qy, qx, (ky * 4 + kx) * 3:(ky * 4 + kx) * 3 + 3] += gn[:, py, px, :]
    torch.cuda.synchronize()
    assert rel(colg, refc) < 5e-3
    z, zt = bf(rng.normal(size=(2, 4, 6, 64)))
    bias = torch.tensor(rng.normal(size=3), dtype=torch.float32).cuda()
    out = K.tap_gather_up2(zt, bias, 4, 1, 3, True)
    refo = np.zeros((2, 8, 12, 3))
    zn = z.numpy()
    for py in range(8):
        for px in range(12):
            for ky in range(4):
                for kx in range(4):
                    iy, ix = py + ky - 1, px + kx - 1
                    if 0 <= iy < 8 and 0 <= ix < 12:
                        refo[:, py, px, :] += zn[:, iy >> 1, ix >> 1, (ky * 4 + kx) * 3:(ky * 4 + kx) * 3 + 3]
    refo = np.tanh(refo + bias.cpu().numpy().astype(np.float64))
    torch.cuda.synchronize()
    assert rel(out, refo) < 5e-3
    # <scatter(g), z> == <g, gather-without-bias(z)>: the two kernels are adjoint
    out_lin = K.tap_gather_up2(zt, None, 4, 1, 3, False).double().cpu().numpy()
    assert abs(float((refc * zn).sum()) - float((gn * out_lin).sum())) < 2e-2 * float(np.abs(refc * zn).sum())
