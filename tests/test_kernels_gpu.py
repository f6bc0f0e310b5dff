"""HIP kernels (through the C ABI) vs the float64 oracle, on a real MI355X.

Tolerances (stated per the bf16 compute path; the reference arithmetic is fp32):
  * inputs are drawn in fp32 and ROUNDED TO bf16 first, and the oracle gets the rounded values, so
    the comparison isolates kernel arithmetic (fp32 accumulate) + one bf16 output rounding;
  * bf16 outputs: max|hip-ref| <= 1e-2 * max|ref|  (bf16 has 2^-9 relative rounding; K up to 9216);
  * fp32 outputs from bf16 operands (wgrad, bias grad, statistics): <= 2e-3 * max|ref|;
  * pure fp32 kernels (spectral norm, Adam): <= 1e-5 relative.
"""
import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu

BF_TOL = 1e-2
F32_FROM_BF_TOL = 2e-3


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gan_lib_tensorflow_amd import kernels
    kernels.lib()
    return kernels


def bf(a):
    """fp32 ndarray -> (bf16-rounded float64 ndarray, bf16 cuda tensor)"""
    t = torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16)
    return t.to(torch.float64).numpy(), t.cuda().contiguous()


def f32(a):
    t = torch.tensor(np.asarray(a, np.float32))
    return t.to(torch.float64).numpy(), t.cuda().contiguous()


def relerr(got, ref):
    got = got.detach().to(torch.float64).cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), "non-finite output"
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def test_tr_read_semantics(K):
    """ds_read_b64_tr_b16: lane i of each 16-lane group receives column i of the 4x16 block."""
    out = K.tr_probe("cuda").cpu().numpy().reshape(64, 4)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        np.testing.assert_array_equal(out[lane], [64 * g + 16 * q + i for q in range(4)])


CONV_CASES = [
    # n, h, cin, cout, k, flags-name
    (2, 8, 64, 64, 3, ""),
    (2, 8, 128, 128, 3, "relu"),
    (3, 8, 64, 128, 1, ""),
    (2, 4, 64, 64, 3, "up"),          # input 4x4 -> output 8x8
    (2, 16, 256, 256, 3, ""),         # 128x128-tile config needs >=192 tiles: M=512 -> uses 64x64
    (64, 16, 256, 256, 3, ""),        # M=16384 -> 128x128 tiles
    (2, 8, 256, 3, 3, "tanh"),        # G.Output shape class (Cout=3, padded to 32)
    (2, 8, 3, 128, 3, ""),            # D.Block.1.Conv1 class (Cin=3, packed K)
    (2, 8, 3, 128, 1, ""),            # D.Block.1.Shortcut class
    (5, 1, 300, 128, 1, ""),          # D.Embedding_y linear (K=300 packed)
    (7, 1, 128, 1, 1, ""),            # D.Output linear
    (3, 6, 64, 96, 3, ""),            # non power-of-two spatial size, Cout=96 (3 tiles of 32)
    (2, 16, 256, 3, 3, "tanh"),       # G.Output through the 32-cout patch kernel (W % 16 == 0)
    (3, 32, 64, 3, 3, "relu"),
    (4, 32, 3, 128, 3, ""),           # narrow-input kernel at the real D.Block.1.Conv1 geometry
    (4, 16, 3, 256, 1, ""),           # narrow-input 1x1
    (1, 5, 3, 128, 3, "relu"),        # narrow-input, ragged pixel count (25 pixels)
    (256, 16, 64, 128, 3, "relu"),    # >= 256 blocks of 16x16 patches: the 8-wave patch kernel
    (3, 8, 256, 128, 1, ""),          # 1x1, Cin = 256
    (1, 5, 128, 192, 3, "relu"),      # ragged pixel count, Cout = 3 tiles of 64
]


@pytest.mark.parametrize("n,h,cin,cout,k,mode", CONV_CASES)
def test_conv_fprop(K, n, h, cin, cout, k, mode):
    rng = np.random.default_rng(n * 1000 + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    w, _ = bf(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin))
    b, bt = f32(rng.normal(size=cout))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    wf, _ = K.prep_weights(wt, True, False)
    flags = 0
    xin = x
    H = h
    if "up" in mode:
        flags |= K.IN_UPSAMPLE2X
        xin = R.upsample_nn2x(x)
        H = 2 * h
    if "relu" in mode:
        flags |= K.IN_RELU
        xin = R.relu(xin)
    if "tanh" in mode:
        flags |= K.OUT_TANH
    res, rest = bf(rng.normal(size=(n, H, H, cout)))
    y = K.conv2d_fprop(xt, wf, bt, (H, H), cout, k, flags, 1.0, rest)
    ref = R.conv2d_same(xin, w, b) + res
    if "tanh" in mode:
        ref = np.tanh(ref)
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL


@pytest.mark.parametrize("n,h,cin,cout,k,mode", [
    (2, 8, 64, 64, 3, ""), (2, 8, 128, 256, 3, "mask"), (2, 8, 64, 128, 1, ""), (2, 8, 128, 128, 3, "pool"),
    (2, 8, 256, 3, 3, ""),       # dgrad of G.Output: Cout=3 -> narrow-input kernel with 256 output rows
    (3, 32, 128, 3, 3, "mask"),
    (2, 8, 3, 128, 3, ""),       # dgrad of D.Block.1.Conv1: output 3 channels
    (4, 1, 300, 128, 1, ""), (4, 1, 128, 1, 1, ""),
])
def test_conv_dgrad(K, n, h, cin, cout, k, mode):
    rng = np.random.default_rng(n * 77 + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    w, _ = bf(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cout))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    _, wd = K.prep_weights(wt, False, True)
    if mode == "pool":   # conv followed by 2x2 mean: dy arrives at half size
        dy, dyt = bf(rng.normal(size=(n, h // 2, h // 2, cout)))
        dx = K.conv2d_dgrad(dyt, wd, (h, h), cin, k, K.IN_UPSAMPLE2X, 0.25)
        ref, _, _ = R.conv2d_same_grads(x, w, R.meanpool2x2_grad(dy))
    else:
        dy, dyt = bf(rng.normal(size=(n, h, h, cout)))
        dx = K.conv2d_dgrad(dyt, wd, (h, h), cin, k, 0, 1.0, None, xt if mode == "mask" else None)
        ref, _, _ = R.conv2d_same_grads(x, w, dy)
        if mode == "mask":
            ref = ref * (x > 0)
    torch.cuda.synchronize()
    assert relerr(dx, ref) < BF_TOL


@pytest.mark.parametrize("n,h,cin,cout,k,mode", [
    (2, 8, 64, 64, 3, ""), (4, 8, 128, 128, 3, "relu"), (2, 8, 128, 256, 1, ""), (2, 4, 64, 64, 3, "up"),
    (2, 8, 64, 128, 3, "pool"), (64, 16, 256, 256, 3, ""),
    (2, 8, 3, 128, 3, ""), (2, 8, 3, 128, 1, ""), (2, 8, 256, 3, 3, ""),
    (6, 1, 300, 128, 1, ""), (9, 1, 128, 1, 1, ""), (3, 6, 64, 96, 3, ""),
    (2, 8, 3, 512, 1, ""), (2, 8, 3, 256, 3, ""),            # narrow input, several 128-channel tiles (PGGAN fromRGB): every tile owns its bias columns
])
def test_conv_wgrad(K, n, h, cin, cout, k, mode):
    rng = np.random.default_rng(n * 31 + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    flags, scale, xin, H = 0, 1.0, x, h
    if mode == "up":
        flags, xin, H = K.IN_UPSAMPLE2X, R.upsample_nn2x(x), 2 * h
    if mode == "relu":
        flags, xin = K.IN_RELU, R.relu(x)
    if mode == "pool":
        dy, dyt = bf(rng.normal(size=(n, h // 2, h // 2, cout)))
        flags, scale = K.DY_UPSAMPLE2X, 0.25
        dyfull = R.meanpool2x2_grad(dy)
    else:
        dy, dyt = bf(rng.normal(size=(n, H, H, cout)))
        dyfull = dy
    dw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device="cuda")
    dbf = torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.conv2d_wgrad(xt, dyt, dw, (H, H), k, flags, scale, dbias=dbf)      # fused bias gradient
    torch.cuda.synchronize()
    assert relerr(dbf, dy.sum(axis=(0, 1, 2))) < F32_FROM_BF_TOL
    w0 = np.zeros((k, k, cin, cout))
    _, ref, refb = R.conv2d_same_grads(xin, w0, dyfull)
    torch.cuda.synchronize()
    assert relerr(dw, ref) < F32_FROM_BF_TOL
    K.conv2d_wgrad(xt, dyt, dw, (H, H), k, flags, scale)     # accumulates
    torch.cuda.synchronize()
    assert relerr(dw, 2 * ref) < F32_FROM_BF_TOL
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.colsum(dyt, db, 1.0)
    torch.cuda.synchronize()
    assert relerr(db, dy.sum(axis=(0, 1, 2))) < F32_FROM_BF_TOL


@pytest.mark.parametrize("k,cin,cout", [(4, 64, 64), (3, 64, 32), (5, 128, 64), (4, 3, 64)])
def test_deconv2d(K, k, cin, cout):
    rng = np.random.default_rng(k + cin)
    n, h = 2, 4
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    f, _ = bf(rng.normal(size=(k, k, cout, cin)) / np.sqrt(k * k * cin))
    b, bt = f32(rng.normal(size=cout))
    ft = torch.tensor(f, dtype=torch.float32).cuda()
    wfz, wz = K.prep_weights(ft, True, True)
    y = K.deconv2d_fprop(xt, wz, bt, cout, k)
    torch.cuda.synchronize()
    assert relerr(y, R.deconv2d_same(x, f, b)) < BF_TOL
    dy, dyt = bf(rng.normal(size=(n, 2 * h, 2 * h, cout)))
    rdx, rdf, _ = R.deconv2d_same_grads(x, f, dy)
    dx = K.deconv2d_dgrad(dyt, wfz, cin, k)
    df = torch.zeros((k, k, cout, cin), dtype=torch.float32, device="cuda")
    K.deconv2d_wgrad(xt, dyt, df, k)
    torch.cuda.synchronize()
    assert relerr(dx, rdx) < BF_TOL
    assert relerr(df, rdf) < F32_FROM_BF_TOL


@pytest.mark.parametrize("n,h,k,cin,cout", [(2, 4, 3, 64, 64), (2, 4, 4, 64, 96), (3, 8, 4, 128, 3), (64, 16, 4, 256, 256), (64, 16, 3, 256, 256), (2, 6, 3, 64, 32)])
def test_deconv2d_phase_form(K, n, h, k, cin, cout):
    """Deconv2D fprop by output phase (k = 3, 4): four 2x2-tap convolutions over the low-resolution input instead of a
    k x k conv over the zero-inserted one, against the oracle's conv2d_transpose and against the zero-insertion path;
    including a realistic shape (n = 64, 16 -> 32, 256 channels: the generator-block size of the SNGAN path) and the
    reference-shaped op end to end (values + the three gradients through autograd)."""
    rng = np.random.default_rng(k * 100 + cin + n)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    f, _ = bf(rng.normal(size=(k, k, cout, cin)) / np.sqrt(k * k * cin / 4))
    b, bt = f32(rng.normal(size=cout))
    ft = torch.tensor(f, dtype=torch.float32).cuda()
    y = K.upconv3x3_fprop(xt, K.deconv2d_prep_phases(ft), bt, cout)
    torch.cuda.synchronize()
    big = n * h * h * cin > 1 << 20
    if not big:
        assert relerr(y, R.deconv2d_same(x, f, b)) < BF_TOL
    _, wz = K.prep_weights(ft, False, True)
    y0 = K.deconv2d_fprop(xt, wz, bt, cout, k)                          # zero-insertion form: same products, other order
    assert relerr(y, y0.double().cpu().numpy()) < (2e-2 if big else 1e-2)
    if not big:
        from gan_lib_tensorflow_amd.common.ops import deconv2d as D
        from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
        store = set_default_store(ParamStore("cuda", seed=1))
        xt2 = xt.clone().requires_grad_(True)
        out = D.Deconv2D(xt2, cin, cout, k, name='T')
        with torch.no_grad():
            store.vars['T/Filters'].copy_(ft)
            store.vars['T/Biases'].copy_(bt)
        out = D.Deconv2D(xt2, cin, cout, k, name='T')
        dy, dyt = bf(rng.normal(size=(n, 2 * h, 2 * h, cout)))
        out.backward(dyt)
        rdx, rdf, rdb = R.deconv2d_same_grads(x, f, dy)
        assert relerr(out, R.deconv2d_same(x, f, b)) < BF_TOL and relerr(xt2.grad, rdx) < BF_TOL
        assert relerr(store.vars['T/Filters'].grad, rdf) < F32_FROM_BF_TOL and relerr(store.vars['T/Biases'].grad, rdb) < F32_FROM_BF_TOL


def test_spectral_norm_batched_fwd_bwd(K):
    rng = np.random.default_rng(5)
    shapes = [(3, 128), (27, 128), (1152, 128), (300, 128), (2304, 256), (128, 1), (70, 33)]
    Ws, us, Gs = [], [], []
    for kk, c in shapes:
        Ws.append(f32(rng.normal(size=(kk, c)) * 0.05))
        us.append(f32(rng.normal(size=(1, c))))
        Gs.append(f32(rng.normal(size=(kk, c))))
    batch = K.SnBatch([w[1] for w in Ws], [u[1] for u in us])
    Wbars = batch.forward()
    dWs = [torch.zeros_like(w[1]) for w in Ws]
    batch.backward([g[1] for g in Gs], dWs)
    torch.cuda.synchronize()
    for i, (w, u, g) in enumerate(zip(Ws, us, Gs)):
        Wb, u1, sigma, v = R.sn_forward(w[0], u[0])
        assert relerr(Wbars[i], Wb) < 1e-5
        assert relerr(batch.u_out_views()[i], u1.ravel()) < 1e-5
        assert abs(float(batch.sigma(i)) - sigma) / sigma < 1e-5
        assert relerr(dWs[i], R.sn_backward(w[0], u[0], g[0])) < 2e-4


@pytest.mark.parametrize("n,hw,c,groups,relu", [(8, 16, 1024, 2, True), (4, 64, 256, 1, True), (6, 256, 256, 2, False), (64, 1024, 256, 2, True)])
def test_cond_batchnorm_fwd_bwd(K, n, hw, c, groups, relu):
    rng = np.random.default_rng(n + hw)
    side = int(round(hw ** 0.5))
    x, xt = bf(rng.normal(size=(n, side, side, c)) * 1.5 + 0.3)
    labels = rng.integers(0, 10, n)
    lt = torch.tensor(labels, dtype=torch.int32).cuda()
    gamma, gt = f32(1 + 0.3 * rng.normal(size=(10, c)))
    beta, bt = f32(0.2 * rng.normal(size=(10, c)))
    y, stats = K.cbn_fwd(xt, lt, gt, bt, groups, relu)
    ry, cache = R.cond_batchnorm_forward(x, labels, gamma, beta, groups)
    ref = R.relu(ry) if relu else ry
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL
    assert relerr(stats[:, 0, :], cache[2].reshape(groups, c)) < 1e-4
    assert relerr(stats[:, 1, :], cache[1].reshape(groups, c)) < 1e-4
    dy, dyt = bf(rng.normal(size=x.shape))
    dg = torch.zeros_like(gt)
    db = torch.zeros_like(bt)
    dx = K.cbn_bwd(dyt, xt, y, lt, gt, stats, dg, db, groups, relu)
    # the kernel masks by ITS bf16 y>0; use the same mask so the comparison is about the arithmetic
    mask = (y.to(torch.float64).cpu().numpy() > 0) if relu else 1.0
    rdx, rdg, rdb = R.cond_batchnorm_backward(dy * mask, labels, gamma, cache, groups)
    torch.cuda.synchronize()
    assert relerr(dx, rdx) < BF_TOL
    assert relerr(dg, rdg) < F32_FROM_BF_TOL
    assert relerr(db, rdb) < F32_FROM_BF_TOL
    # the mask recomputed from x (the forward pass's expression in the forward pass's order) instead of read from y:
    # the same input gradient and table gradients up to fp32 summation order
    dg2, db2 = torch.zeros_like(gt), torch.zeros_like(bt)
    dx2 = K.cbn_bwd(dyt, xt, None, lt, gt, stats, dg2, db2, groups, relu, beta=bt)
    torch.cuda.synchronize()
    assert relerr(dx2, dx.double().cpu().numpy()) < 4e-3            # (the mask of a value at the rounding boundary of y may differ)
    assert relerr(dg2, dg.double().cpu().numpy()) < 1e-5 and relerr(db2, db.double().cpu().numpy()) < 1e-5
    # gank_cbn_bwd_ws on the large workspace (the default above): the pixel parts of a sample in rows of their own, added in a fixed
    # order -- the same bits every run; the older entries (fill launch + fp32 atomics) agree up to summation order
    dg3, db3 = torch.zeros_like(gt), torch.zeros_like(bt)
    dx3 = K.cbn_bwd(dyt, xt, y, lt, gt, stats, dg3, db3, groups, relu)
    assert torch.equal(dx3, dx) and torch.equal(dg3, dg) and torch.equal(db3, db)
    K.CBN_BWD_PART_ROWS = False
    try:
        dg4, db4 = torch.zeros_like(gt), torch.zeros_like(bt)
        dx4 = K.cbn_bwd(dyt, xt, y, lt, gt, stats, dg4, db4, groups, relu)
        dg5, db5 = torch.zeros_like(gt), torch.zeros_like(bt)
        dx5 = K.cbn_bwd(dyt, xt, None, lt, gt, stats, dg5, db5, groups, relu, beta=bt)
    finally:
        K.CBN_BWD_PART_ROWS = True
    torch.cuda.synchronize()
    assert relerr(dx4, rdx) < BF_TOL and relerr(dg4, dg.double().cpu().numpy()) < 1e-5 and relerr(db4, db.double().cpu().numpy()) < 1e-5
    assert relerr(dx5, dx2.double().cpu().numpy()) < 4e-3 and relerr(dg5, dg2.double().cpu().numpy()) < 1e-5


def test_pool_unpool_add_relu_tanh(K):
    rng = np.random.default_rng(11)
    for c in (128, 3):
        x, xt = bf(rng.normal(size=(3, 8, 8, c)))
        r, rt = bf(rng.normal(size=(3, 4, 4, c)))
        assert relerr(K.pool2x2(xt, 0.25, rt), R.meanpool2x2(x) + r) < BF_TOL
        assert relerr(K.pool2x2(xt, 1.0), 4 * R.meanpool2x2(x)) < BF_TOL
        assert relerr(K.unpool2x2_add(rt, xt, 0.25), x + R.meanpool2x2_grad(r)) < BF_TOL
        assert relerr(K.unpool2x2_add(rt, None, 1.0), R.upsample_nn2x(r)) == 0.0
    a, at = bf(rng.normal(size=1003))
    b, bt = bf(rng.normal(size=1003))
    assert relerr(K.add(at, bt), a + b) < BF_TOL
    assert relerr(K.relu_fwd(at), R.relu(a)) == 0.0
    assert relerr(K.relu_fwd(at, 0.2), R.lrelu(a)) < BF_TOL
    assert relerr(K.relu_bwd(bt, at), b * (a > 0)) == 0.0
    y = np.tanh(a)
    yq, yt = bf(y)
    assert relerr(K.tanh_bwd(bt, yt), b * (1 - yq * yq)) < BF_TOL
    f, ft = f32(rng.normal(size=77))
    assert relerr(K.to_f32(K.to_bf16(ft)), bf(f)[0]) == 0.0


def test_relu_meanpool_concat_embedding(K):
    rng = np.random.default_rng(12)
    x, xt = bf(rng.normal(size=(5, 8, 8, 128)))
    y = K.relu_meanpool_hw_fwd(xt)
    assert relerr(y, R.relu_mean_hw(x)) < BF_TOL
    dy, dyt = bf(rng.normal(size=(5, 128)))
    assert relerr(K.relu_meanpool_hw_bwd(dyt, xt), (x > 0) * dy[:, None, None, :] / 64.) < BF_TOL
    a, at = bf(rng.normal(size=(3, 4, 4, 128)))
    e, et = bf(rng.normal(size=(3, 128)))
    cat = K.concat_tile_fwd(at, et)
    ref = np.concatenate([a, np.broadcast_to(e[:, None, None, :], (3, 4, 4, 128))], axis=3)
    assert relerr(cat, ref) == 0.0
    g, gt = bf(rng.normal(size=(3, 4, 4, 256)))
    da, de = K.concat_tile_bwd(gt, 128)
    assert relerr(da, g[..., :128]) == 0.0
    assert relerr(de, g[..., 128:].sum(axis=(1, 2))) < BF_TOL
    table, tt = f32(rng.uniform(-0.08, 0.08, size=(10, 300)))
    idx = rng.integers(0, 10, 17)
    it = torch.tensor(idx, dtype=torch.int32).cuda()
    emb = K.embedding_fwd(tt, it)
    assert relerr(emb, bf(table[idx])[0]) == 0.0
    d, dt = bf(rng.normal(size=(17, 300)))
    dtab = torch.zeros_like(tt)
    K.embedding_bwd(dt, it, dtab)
    assert relerr(dtab, R.embed_y_grad(idx, d, 10)) < 1e-5


def test_losses(K):
    rng = np.random.default_rng(13)
    l, lt = bf(rng.normal(size=128) * 2)
    loss, dl, dl32 = K.hinge_d_loss(lt, 64)
    rl, rd = R.hinge_d_loss(l, 64)
    assert abs(float(loss) - rl) < 1e-5 and relerr(dl, rd) < BF_TOL and relerr(dl32, rd) < 1e-6
    loss, dl, dl32 = K.hinge_g_loss(lt)
    rl, rd = R.hinge_g_loss(l)
    assert abs(float(loss) - rl) < 1e-5 and relerr(dl, rd) < BF_TOL and relerr(dl32, rd) < 1e-6
    lg, lgt = bf(rng.normal(size=(32, 10)) * 3)
    lb = rng.integers(0, 10, 32)
    loss, dl, dl32 = K.softmax_xent(lgt, torch.tensor(lb, dtype=torch.int32).cuda())
    rl, rd = R.softmax_xent(lg, lb)
    assert abs(float(loss) - rl) < 1e-4 and relerr(dl, rd) < BF_TOL and relerr(dl32, rd) < 1e-5


def test_scaled_and_summed_losses_scale_their_gradients(K):
    """gen_cost + ACGAN_SCALE_G * xent (gan_cifar_resnet.py:476), loss / accum_steps, a non-power-of-two batch (100):
    the upstream gradient reaches the logits (_Loss.backward used to drop it)."""
    from gan_lib_tensorflow_amd import functional as Fn
    rng = np.random.default_rng(113)
    lg, lgt = bf(rng.normal(size=(100, 10)) * 3)
    lb = rng.integers(0, 10, 100)
    lbt = torch.tensor(lb, dtype=torch.int32).cuda()
    l, lt = bf(rng.normal(size=100) * 2)
    lgt.requires_grad_(True)
    lt.requires_grad_(True)
    total = Fn.hinge_g_loss(lt) + 0.1 * Fn.softmax_xent(lgt, lbt)
    total.backward()
    _, rdx = R.softmax_xent(lg, lb)
    _, rdg = R.hinge_g_loss(l)
    assert relerr(lgt.grad, 0.1 * rdx) < BF_TOL
    assert relerr(lt.grad, rdg) < BF_TOL
    # one rounding: the scaled gradient is bf16(fp32 product), not bf16(bf16(1/n) * 0.5)
    lt.grad = None
    (0.5 * Fn.hinge_d_loss(lt, 50)).backward()
    _, rdd = R.hinge_d_loss(l, 50)
    want = torch.tensor(0.5 * rdd, dtype=torch.float32).to(torch.bfloat16)
    assert torch.equal(lt.grad.cpu(), want)
    # the train step's unit seed takes the no-launch path and gives the unscaled gradient
    lt.grad = None
    loss = Fn.hinge_d_loss(lt, 50)
    loss.backward(gradient=Fn.unit_seed(loss))
    assert relerr(lt.grad, rdd) < BF_TOL


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_critic_head_equals_linear_plus_hinge(K, mode):
    """gank_critic_head_hinge: D.Output + hinge loss + the layer's three gradients in one launch == linear_fwd, hinge_*_loss,
    linear_bwd one after the other (bit-for-bit for the logits, the loss and d loss / d x; weight gradient to fp32 summation
    order), and matches the oracle."""
    rng = np.random.default_rng(21 + mode)
    m, k, n_real = 128, 128, 64
    x, xt = bf(rng.normal(size=(m, k)))
    w, wt = f32(rng.normal(size=(k, 1)) * 0.2)
    b, bt = f32(rng.normal(size=1))
    gw = torch.full((k,), 0.5, dtype=torch.float32, device="cuda")          # accumulates on top of existing content
    gb = torch.full((1,), 0.25, dtype=torch.float32, device="cuda")
    loss, logits, dx = K.critic_head_hinge(xt, wt.view(-1), bt, n_real, mode, True, gw, gb)
    y = K.linear_fwd(xt, wt, bt)
    l0, dl, _ = K.hinge_d_loss(y.view(-1), n_real) if mode == 0 else K.hinge_g_loss(y.view(-1))
    gw0 = torch.zeros((k, 1), dtype=torch.float32, device="cuda")
    gb0 = torch.zeros(1, dtype=torch.float32, device="cuda")
    dx0 = K.linear_bwd(dl.view(m, 1), xt, wt, True, gw0, gb0)
    torch.cuda.synchronize()
    assert torch.equal(logits, y.view(-1)) and torch.equal(dx, dx0)
    assert abs(float(loss) - float(l0)) < 1e-6
    assert relerr(gw - 0.5, gw0.view(-1).double().cpu().numpy()) < 1e-5 and abs(float(gb) - 0.25 - float(gb0)) < 1e-6
    lg = x @ w[:, 0] + b[0]
    rl, rd = R.hinge_d_loss(lg, n_real) if mode == 0 else R.hinge_g_loss(lg)
    assert abs(float(loss) - rl) < 2e-2 * max(1.0, abs(rl))
    # through autograd, with and without the weight gradients
    from gan_lib_tensorflow_amd import functional as Fn
    xg = xt.clone().requires_grad_(True)
    wg, bg = wt.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    out = Fn.hinge_d_head(xg, wg, bg, n_real) if mode == 0 else Fn.hinge_g_head(xg, wg, bg)
    out.backward(gradient=Fn.unit_seed(out))
    torch.cuda.synchronize()
    assert torch.equal(out.logits, logits) and torch.equal(xg.grad, dx0)
    assert relerr(wg.grad, gw0.double().cpu().numpy()) < 1e-5 and relerr(bg.grad, gb0.double().cpu().numpy()) < 1e-5
    with pytest.raises(NotImplementedError):
        (0.5 * (Fn.hinge_g_head(xg, wg, bg))).backward()                      # a weighted loss goes through the unfused operators


@pytest.mark.parametrize("n,h,cout,groups", [(8, 32, 3, 2), (6, 16, 128, 3)])
def test_cbn_relu_fused_into_conv_operand_staging(K, n, h, cout, groups):
    """gank_cbn_relu_conv3x3_fprop: conv3x3(relu(cond_batchnorm(x))) with the normalisation applied while the conv stages its
    operand == the two launches one after the other, bit for bit (same expression, same order, same bf16 rounding of the
    normalised value), with statistics from the statistics pass or from a conv epilogue's sums."""
    rng = np.random.default_rng(n + h + cout)
    c = 256
    x, xt = bf(rng.normal(size=(n, h, h, c)) * 1.7 + 0.3)
    labels = torch.tensor(rng.integers(0, 10, n), dtype=torch.int32).cuda()
    gamma = torch.tensor(rng.normal(size=(10, c)) * 0.2 + 1, dtype=torch.float32).cuda()
    beta = torch.tensor(rng.normal(size=(10, c)) * 0.2, dtype=torch.float32).cuda()
    w, _ = bf(rng.normal(size=(3, 3, c, cout)) / np.sqrt(9 * c))
    b, bt = f32(rng.normal(size=cout))
    wf, _ = K.prep_weights(torch.tensor(w, dtype=torch.float32).cuda(), True, False)
    flags = K.OUT_TANH if cout == 3 else 0
    yn, stats0 = K.cbn_fwd(xt, labels, gamma, beta, groups, True)
    ref = K.conv2d_fprop(yn, wf, bt, (h, h), cout, 3, flags)
    stats = K.cbn_stats(xt, groups)
    y = K.cbn_relu_conv3x3_fprop(xt, labels, gamma, beta, stats, wf, bt, cout, flags)
    torch.cuda.synchronize()
    assert torch.equal(stats, stats0) and torch.equal(y, ref)
    # against the oracle too
    ry, _ = R.cond_batchnorm_forward(x, labels.cpu().numpy(), gamma.double().cpu().numpy(), beta.double().cpu().numpy(), groups)
    rr = R.conv2d_same(R.relu(ry), w, b)
    assert relerr(y, np.tanh(rr) if cout == 3 else rr) < 2 * BF_TOL


def test_adam_tf_and_lr_decay(K):
    rng = np.random.default_rng(14)
    n = 1003
    p0, pt = f32(rng.normal(size=n))
    m = torch.zeros(n, dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.float32, device="cuda")
    hp = torch.tensor([2e-4, 0., 0.9, 1e-8, 0.5, 1.0, 0, 0], dtype=torch.float32, device="cuda")
    t = torch.zeros(1, dtype=torch.int64, device="cuda")
    it = torch.tensor([30000], dtype=torch.int64, device="cuda")
    rp, rm, rv = p0.copy(), np.zeros(n), np.zeros(n)
    for step in range(1, 4):
        g, gt = f32(rng.normal(size=n))
        K.adam_tf(pt, gt, m, v, hp, t, it)
        rp, rm, rv = R.adam_tf_step(rp, 0.5 * g, rm, rv, step, 2e-4 * R.lr_decay(30000))
    torch.cuda.synchronize()
    assert int(t) == 3
    assert relerr(pt, rp) < 1e-5 and relerr(v, rv) < 1e-5
    assert int(hp.view(torch.int32)[6]) == 0                       # the ticket word is back at zero after every launch
    # zero_grads: the whole gradient buffer (it may carry a scratch tail behind the gradients) is cleared by the same launch
    n2 = 4096
    p2 = torch.zeros(n2, dtype=torch.float32, device="cuda")
    g2 = torch.ones(2 * n2, dtype=torch.float32, device="cuda")
    m2, v2 = torch.zeros_like(p2), torch.zeros_like(p2)
    K.adam_tf(p2, g2, m2, v2, hp, t, it, zero_grads=True)
    torch.cuda.synchronize()
    assert int(t) == 4 and float(g2.abs().max()) == 0.0 and float(p2.abs().min()) > 0.0


def test_generator_feed_is_the_three_launches_in_front_of_a_generator_pass(K):
    """gank_generator_feed (gan_cifar_resnet.py:240,467): label draw + noise draw + statistics-arena fill in one launch -- the values
    and the stream offset of gank_rng_labels, gank_rng_normal_bf16 and a zero fill, with and without the label draw."""
    for n_lab in (128, 0, 5):
        a, b = K.new_rng_state(321, "cuda"), K.new_rng_state(321, "cuda")
        for _ in range(2):          # two passes in a row: the offsets keep agreeing
            lab_ref = K.rng_labels(n_lab, 10, a) if n_lab else None
            z_ref = K.rng_normal((max(n_lab, 3), 128), a)
            lab, z, zb = K.generator_feed(b, (max(n_lab, 3), 128), 1027 * 4, n_lab, 10)
            torch.cuda.synchronize()
            assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16)) and torch.equal(a, b)
            assert (lab is None) == (n_lab == 0) and (n_lab == 0 or torch.equal(lab, lab_ref))
            assert zb.shape == (1027 * 4,) and float(zb.abs().max()) == 0.0
    zb = torch.full((4 * 37 + 4,), 3.0, device="cuda")      # a fill that is not a multiple of four floats
    _lib_fill = K.lib().gank_generator_feed
    st = K.new_rng_state(1, "cuda")
    z = torch.empty(8, dtype=torch.bfloat16, device="cuda")
    assert _lib_fill(None, 0, 10, z.data_ptr(), 8, zb.data_ptr(), 4 * 37 + 3, st.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert float(zb[:4 * 37 + 3].abs().max()) == 0.0 and float(zb[4 * 37 + 3]) == 3.0


def test_preprocess_and_rng(K):
    st = K.new_rng_state(42, "cuda")
    data = torch.randint(0, 256, (16, 3072), dtype=torch.uint8)
    y = K.preprocess_real(data.cuda(), st).to(torch.float64).cpu().numpy().reshape(16, 3072)
    base = R.preprocess_real(data.numpy(), np.zeros((16, 3072)))
    d = y - base
    # dequantisation noise U[0,1/128) then bf16 rounding (|x|<=1 -> ulp <= 2^-8)
    assert d.min() > -2 ** -8 and d.max() < 1 / 128 + 2 ** -8
    assert int(st[1]) == 1
    z = K.rng_normal((64, 128), st).to(torch.float64).cpu().numpy()
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05 and int(st[1]) == 2
    z2 = K.rng_normal((64, 128), st).to(torch.float64).cpu().numpy()
    assert np.abs(z - z2).max() > 0.1            # the offset advanced: fresh numbers
    lb = K.rng_labels(4096, 10, st).cpu().numpy()
    assert lb.min() == 0 and lb.max() == 9 and np.bincount(lb).min() > 300


@pytest.mark.parametrize("n,hl,cin,cout", [(2, 4, 64, 64), (3, 8, 128, 256), (64, 16, 256, 256), (2, 8, 256, 32)])
def test_upconv3x3_phase_decomposition(K, n, hl, cin, cout):
    """NN-upsample + 3x3 SAME conv (gan_cifar_resnet.py:140-153) computed as the 4 output phases of the
    equivalent 4x4 stride-2 transposed conv, and its input gradient as the 4x4 stride-2 conv of dy."""
    rng = np.random.default_rng(n + hl + cin)
    x, xt = bf(rng.normal(size=(n, hl, hl, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout))
    res, rest = bf(rng.normal(size=(n, 2 * hl, 2 * hl, cout)))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    wph, wd4 = K.upconv3x3_prep(wt)
    y = K.upconv3x3_fprop(xt, wph, bt, cout, 0, rest)
    ref = R.conv2d_same(R.upsample_nn2x(x), w, b) + res
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL
    dy, dyt = bf(rng.normal(size=(n, 2 * hl, 2 * hl, cout)))
    dx = K.upconv3x3_dgrad(dyt, wd4, cin)
    dx_hi, _, _ = R.conv2d_same_grads(R.upsample_nn2x(x), w, dy)
    torch.cuda.synchronize()
    assert relerr(dx, R.upsample_nn2x_grad(dx_hi)) < BF_TOL


@pytest.mark.parametrize("n,hl,cin,cout", [(2, 8, 64, 64), (3, 8, 128, 64), (4, 16, 256, 256), (128, 16, 256, 256), (128, 4, 1024, 256)])
def test_upconv3x3_filter_gradient_phase_form(K, n, hl, cin, cout):
    """gank_upconv3x3_wgrad: the filter gradient of NN-upsample + 3x3 SAME conv (gan_cifar_resnet.py:138-153) as 16 (phase, tap)
    products at LOW resolution -- the ConvMeanPool rows kernel with its operands swapped -- folded onto the 3 x 3 taps, against
    the float64 gradient through the explicit upsample (small cases: <= 2e-3 of the maximum, fp32 sums of bf16 products) and
    against the all-taps kernel on the upsampled input (every case: the two sum the same products in different orders, <= 1e-3);
    it ACCUMULATES into dw.  Shapes the rows kernel does not serve (4 x 4 inputs) report a workspace of 0."""
    rng = np.random.default_rng(3000 + n + hl + cin)
    x, xt = bf(rng.normal(size=(n, hl, hl, cin)))
    dy, dyt = bf(rng.normal(size=(n, 2 * hl, 2 * hl, cout)))
    if K.upconv3x3_wgrad_ws(xt, cout) == 0:
        assert hl < 8                                   # the rows kernel wants >= 8 x 8 low-resolution images
        with pytest.raises(RuntimeError):
            K.upconv3x3_wgrad(xt, dyt, torch.zeros((3, 3, cin, cout), device="cuda"))
        return
    dw = torch.full((3, 3, cin, cout), 0.25, device="cuda")
    K.upconv3x3_wgrad(xt, dyt, dw)
    ref = torch.zeros((3, 3, cin, cout), device="cuda")
    K.conv2d_wgrad(xt, dyt, ref, (2 * hl, 2 * hl), 3, K.IN_UPSAMPLE2X)
    torch.cuda.synchronize()
    got = (dw - 0.25).double().cpu().numpy()
    assert relerr(got, ref.double().cpu().numpy()) < 1e-3
    if n <= 4:
        w0 = np.zeros((3, 3, cin, cout))
        _, dw_ref, _ = R.conv2d_same_grads(R.upsample_nn2x(x), w0, dy)
        assert relerr(got, dw_ref) < 2e-3


def test_copy_bytes(K):
    """Kernel-based device copy (the u.assign / feed copies of the captured step): bit-exact, any size/alignment."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for nbytes, so, do in ((1, 0, 0), (15, 0, 0), (16, 0, 0), (4099, 0, 0), (4099, 3, 0), (4099, 16, 5), (1 << 20, 0, 0), ((1 << 20) + 7, 16, 16)):
        src = torch.randint(0, 256, (nbytes + 64,), generator=g, dtype=torch.uint8).cuda()
        dst = torch.full((nbytes + 64,), 7, dtype=torch.uint8, device="cuda")
        K.copy_(dst[do:do + nbytes], src[so:so + nbytes])
        torch.cuda.synchronize()
        assert torch.equal(dst[do:do + nbytes], src[so:so + nbytes])
        assert bool((dst[:do] == 7).all()) and bool((dst[do + nbytes:] == 7).all())     # nothing outside the range
    a = torch.randn(1000, generator=g).cuda()
    assert torch.equal(K.clone(a), a)


@pytest.mark.parametrize("n,hp,cin,cout,relu", [(2, 4, 64, 64, False), (3, 8, 128, 128, True), (128, 16, 128, 128, True),
                                                (16, 8, 256, 128, True), (2, 4, 128, 64, False)])
def test_convpool3x3_as_stride2_conv(K, n, hp, cin, cout, relu):
    """ConvMeanPool 3x3 (gan_cifar_resnet.py:112-123): mean_pool(conv3x3(relu?(x)) + b) + residual run as ONE 4x4
    stride-2 conv; its input gradient as the 4-phase transposed conv, its filter gradient folded from 16 taps."""
    rng = np.random.default_rng(n * 7 + hp + cin)
    x, xt = bf(rng.normal(size=(n, 2 * hp, 2 * hp, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout))
    res, rest = bf(rng.normal(size=(n, hp, hp, cout)))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    wp4, wphd = K.convpool3x3_prep(wt)
    xin = R.relu(x) if relu else x
    y = K.convpool3x3_fprop(xt, wp4, bt, cout, K.IN_RELU if relu else 0, rest)
    ref = R.meanpool2x2(R.conv2d_same(xin, w, b)) + res
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL
    dy, dyt = bf(rng.normal(size=(n, hp, hp, cout)))
    dx_ref, dw_ref, db_ref = R.conv2d_same_grads(xin, w, R.meanpool2x2_grad(dy))
    if relu:
        dx_ref = dx_ref * (x > 0)
    dx = K.convpool3x3_dgrad(dyt, wphd, cin, xt if relu else None)
    torch.cuda.synchronize()
    assert relerr(dx, dx_ref) < BF_TOL
    dw = torch.zeros((3, 3, cin, cout), dtype=torch.float32, device="cuda")
    dw += 1.0                                  # accumulates on top of existing content
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.convpool3x3_wgrad(xt, dyt, dw, K.IN_RELU if relu else 0, dbias=db)
    torch.cuda.synchronize()
    assert relerr(dw - 1.0, dw_ref) < F32_FROM_BF_TOL
    assert relerr(db, dy.sum((0, 1, 2))) < F32_FROM_BF_TOL


@pytest.mark.parametrize("m,k,c", [(128, 300, 128), (128, 128, 1), (5, 7, 3), (64, 128, 10), (33, 65, 130)])
def test_linear_small(K, m, k, c):
    """tf.matmul + bias_add on the fp32 weight (linear.py:161-180) and its three gradients."""
    rng = np.random.default_rng(m + k + c)
    x, xt = bf(rng.normal(size=(m, k)))
    w, wt = f32(rng.normal(size=(k, c)) / np.sqrt(k))
    b, bt = f32(rng.normal(size=c))
    y = K.linear_fwd(xt, wt, bt)
    torch.cuda.synchronize()
    assert relerr(y, R.linear(x, w, b)) < BF_TOL
    dy, dyt = bf(rng.normal(size=(m, c)))
    dw = torch.full((k, c), 0.5, dtype=torch.float32, device="cuda")
    db = torch.full((c,), -1.0, dtype=torch.float32, device="cuda")
    dx = K.linear_bwd(dyt, xt, wt, True, dw, db)
    dx_ref, dw_ref = R.linear_grads(x, w, dy)[:2]
    torch.cuda.synchronize()
    assert relerr(dx, dx_ref) < BF_TOL
    assert relerr(dw - 0.5, dw_ref) < 1e-5
    assert relerr(db + 1.0, dy.sum(0)) < 1e-5
    db2 = torch.zeros(c, dtype=torch.float32, device="cuda")
    assert K.linear_bwd(dyt, None, None, False, None, db2) is None        # bias gradient alone
    torch.cuda.synchronize()
    assert relerr(db2, dy.sum(0)) < 1e-5


def test_batched_prep_kinds_match_standalone(K):
    """One batched launch builds plain, UpsampleConv (kind 1) and ConvMeanPool (kind 2) operands: bit-identical to
    the per-layer entry points, buffers reused in place on the second call."""
    g = torch.Generator(device="cpu").manual_seed(11)
    ws = [torch.randn(s, generator=g).cuda() for s in ((3, 3, 64, 128), (3, 3, 128, 64), (3, 3, 64, 64), (128, 96), (1, 1, 64, 32))]
    kinds = [1, 2, 0, 0, None]
    K.prep_weights_batched(ws, want_d=True, kinds=kinds)
    torch.cuda.synchronize()
    assert not hasattr(ws[4], "_prep")
    ref_up = K.upconv3x3_prep(ws[0].clone())
    ref_pool = K.convpool3x3_prep(ws[1].clone())
    ref_plain = K.prep_weights(ws[2], True, True)
    ref_lin = K.prep_weights(ws[3].view(1, 1, 128, 96), True, True)
    torch.cuda.synchronize()
    for got, ref in ((ws[0]._prep_up, ref_up), (ws[1]._prep_pool, ref_pool), (ws[2]._prep, ref_plain), (ws[3]._prep, ref_lin)):
        for a, b in zip(got, ref):
            assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16))
    ptrs = [t.data_ptr() for t in ws[0]._prep_up + ws[1]._prep_pool + ws[2]._prep]
    ws[0].mul_(2.0)
    K.prep_weights_batched(ws, want_d=True, kinds=kinds)
    torch.cuda.synchronize()
    assert ptrs == [t.data_ptr() for t in ws[0]._prep_up + ws[1]._prep_pool + ws[2]._prep]
    assert torch.equal(ws[0]._prep_up[0].float(), ref_up[0].float() * 2)


def test_critic_feed_matches_separate_launches(K):
    """gank_critic_feed == gank_preprocess_real + the staging copies + the concat, bit for bit, and walks the ring."""
    g = torch.Generator(device="cpu").manual_seed(3)
    b, slots = 8, 3
    real_all = torch.randint(0, 256, (slots, b, 3072), generator=g, dtype=torch.uint8).cuda()
    labels_all = torch.randint(0, 10, (slots, b), generator=g, dtype=torch.int32).cuda()
    fake_all = torch.randn((slots, b, 3072), generator=g).to(torch.bfloat16).cuda()
    both = torch.zeros((2 * b, 3072), dtype=torch.bfloat16, device="cuda")
    labels2 = torch.zeros(2 * b, dtype=torch.int32, device="cuda")
    slot = torch.zeros(1, dtype=torch.int32, device="cuda")
    done = torch.zeros(1, dtype=torch.int32, device="cuda")
    rng_a, rng_b = K.new_rng_state(99, "cuda"), K.new_rng_state(99, "cuda")
    for step in range(slots + 1):
        i = step % slots
        K.critic_feed(real_all, labels_all, fake_all, both, labels2, slot, rng_a, done)
        ref_real = K.preprocess_real(real_all[i], rng_b).reshape(b, 3072)
        torch.cuda.synchronize()
        assert torch.equal(both[:b].view(torch.int16), ref_real.view(torch.int16))
        assert torch.equal(both[b:].view(torch.int16), fake_all[i].view(torch.int16))
        assert torch.equal(labels2[:b], labels_all[i]) and torch.equal(labels2[b:], labels_all[i])
        assert int(slot) == (i + 1) % slots and int(done) == 0 and torch.equal(rng_a, rng_b)


def test_critic_feed_as_a_block_range_of_the_second_spectral_norm_launch(K):
    """gank_sn_power_iter_fwd_b_prep_feed: a deferred critic feed (K.defer_critic_feed) is launched by the next spectral-norm forward
    pass -- inside its second launch when the power iteration has been run already (valid persistent state), as the feed's own
    launch otherwise.  Either way the feed's outputs and counters and the pass's normalised weights, operand copies and u equal
    the two separate launches bit for bit, and the ring walks on."""
    g = torch.Generator(device="cpu").manual_seed(4)
    b, slots = 64, 3
    real_all = torch.randint(0, 256, (slots, b, 3072), generator=g, dtype=torch.uint8).cuda()
    labels_all = torch.randint(0, 10, (slots, b), generator=g, dtype=torch.int32).cuda()
    fake_all = torch.randn((slots, b, 3072), generator=g).to(torch.bfloat16).cuda()
    shapes = [(3, 3, 3, 128), (3, 3, 128, 128), (300, 128), (1, 1, 256, 128), (128, 1)]
    kinds = [0, 4, None, 0, None]

    def build():
        gg = torch.Generator(device="cpu").manual_seed(8)
        Ws = [(torch.randn(sh, generator=gg) * 0.05).cuda() for sh in shapes]
        u_flat = torch.randn(sum(sh[-1] for sh in shapes), generator=gg).cuda()
        us, o = [], 0
        for sh in shapes:
            us.append(u_flat[o:o + sh[-1]].view(1, sh[-1]))
            o += sh[-1]
        return dict(Ws=Ws, us=us, u_flat=u_flat, st=K.SnState(Ws, us, u_flat), both=torch.zeros((2 * b, 3072), dtype=torch.bfloat16, device="cuda"),
                    labels2=torch.zeros(2 * b, dtype=torch.int32, device="cuda"), slot=torch.zeros(1, dtype=torch.int32, device="cuda"),
                    done=torch.zeros(1, dtype=torch.int32, device="cuda"), rng=K.new_rng_state(7, "cuda"))

    A, B = build(), build()
    for step in range(slots + 1):
        valid = step != 1                     # step 1: no power iteration in hand -> the feed's own launch in front of the two-launch pass
        outs = []
        for d, fused in ((A, False), (B, True)):
            if valid:
                d["st"].refresh()
            else:
                d["st"].valid = False
            feed = (real_all, labels_all, fake_all, d["both"], d["labels2"], d["slot"], d["rng"], d["done"])
            batch = K.SnBatch(d["Ws"], d["us"], snapshot=True, inplace=True, state=d["st"])
            batch.prep = (kinds, True)
            if fused:
                K.defer_critic_feed(*feed)
                assert K.deferred_critic_feed_pending()
            else:
                K.critic_feed(*feed)
            outs.append(batch.forward())
            assert not K.deferred_critic_feed_pending()
        torch.cuda.synchronize()
        i = step % slots
        for key in ("both", "labels2", "slot", "done", "rng", "u_flat"):
            assert torch.equal(A[key], B[key]), (step, key)
        assert torch.equal(B["both"][b:].view(torch.int16), fake_all[i].reshape(b, 3072).view(torch.int16)) and int(B["slot"]) == (i + 1) % slots and int(B["done"]) == 0
        for x, y in zip(*outs):
            assert torch.equal(x, y)
            for attr in ("_prep", "_prep_res"):
                pa, pb = getattr(x, attr, None), getattr(y, attr, None)
                assert (pa is None) == (pb is None)
                if pa is not None:
                    for u, v in zip(pa, pb):
                        assert (u is None) == (v is None) and (u is None or torch.equal(u.view(torch.int16), v.view(torch.int16)))


def test_batch_norm_op_and_get_loss(K):
    """a4' / a15 of the scope table: tf.contrib batch_norm in train mode (normalization.py:8-24) with its moving
    statistics, and common/misc.py get_loss('HINGE'), through the reference-shaped ops."""
    from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
    from gan_lib_tensorflow_amd.common.ops import normalization as Nm
    from gan_lib_tensorflow_amd.common import misc
    rng = np.random.default_rng(12)
    n, hw, c = 6, 4, 64
    store = set_default_store(ParamStore("cuda", seed=0))
    moving = dict(moving_mean=np.zeros(c), moving_variance=np.ones(c), biased=np.zeros(c), local_step=0.0)
    gamma = rng.normal(size=c).astype(np.float32) + 1.0
    beta = rng.normal(size=c).astype(np.float32)
    for step in range(2):
        x, xt = bf(rng.normal(size=(n, hw, hw, c)) * 2.0 + 0.5)
        xt.requires_grad_(True)
        with store.variable_scope("D.BN", reuse=step > 0):
            if step == 0:
                y = Nm.batch_norm(xt)                      # creates the variables
                with torch.no_grad():
                    store.vars["D.BN/BatchNorm/gamma"].copy_(torch.tensor(gamma).view(1, c))
                    store.vars["D.BN/BatchNorm/beta"].copy_(torch.tensor(beta).view(1, c))
                    for k in ("moving_mean", "moving_variance", "moving_mean/biased", "moving_mean/local_step"):
                        v = store.vars["D.BN/BatchNorm/" + k]
                        v.copy_(torch.ones_like(v) if k == "moving_variance" else torch.zeros_like(v))
            y = Nm.batch_norm(xt)
        ref, cache, moving = R.batch_norm_train(x, gamma, beta, moving)
        dy, dyt = bf(rng.normal(size=ref.shape))
        for k in ("gamma", "beta"):
            store.vars["D.BN/BatchNorm/" + k].grad = None
        y.backward(dyt)
        dx_ref, dg_ref, db_ref = R.cond_batchnorm_backward(dy, np.zeros(n, np.int64), gamma.reshape(1, c).astype(np.float64), cache, 1)
        torch.cuda.synchronize()
        assert relerr(y, ref) < BF_TOL
        assert relerr(xt.grad, dx_ref) < 2 * BF_TOL
        assert relerr(store.vars["D.BN/BatchNorm/gamma"].grad, dg_ref) < F32_FROM_BF_TOL
        assert relerr(store.vars["D.BN/BatchNorm/beta"].grad, db_ref) < F32_FROM_BF_TOL
        assert relerr(store.vars["D.BN/BatchNorm/moving_mean"], moving["moving_mean"]) < 1e-4
        assert relerr(store.vars["D.BN/BatchNorm/moving_variance"], moving["moving_variance"]) < 1e-3
        assert float(store.vars["D.BN/BatchNorm/moving_mean/local_step"]) == step + 1
    with pytest.raises(NotImplementedError):
        Nm.batch_norm(xt, is_training=False)
    real, realt = bf(rng.normal(size=7))
    fake, faket = bf(rng.normal(size=5))
    realt.requires_grad_(True)
    faket.requires_grad_(True)
    d_loss, g_loss = misc.get_loss(realt, faket, 'HINGE')
    lref, dref = R.hinge_d_loss(np.concatenate([real, fake]), 7)
    gref, _ = R.hinge_g_loss(fake)
    d_loss.backward()
    torch.cuda.synchronize()
    assert abs(float(d_loss) - lref) < 1e-5 and abs(float(g_loss) - gref) < 1e-5
    assert relerr(realt.grad, dref[:7]) < BF_TOL and relerr(faket.grad, dref[7:]) < BF_TOL
    # 'WGAN' / 'WGAN-GP' (misc.py:328-352): -mean(real) + mean(fake); -mean(fake)
    realt.grad = faket.grad = None
    d_loss, g_loss = misc.get_loss(realt, faket, 'WGAN-GP')
    d_loss.backward()
    torch.cuda.synchronize()
    assert abs(float(d_loss) - (-real.mean() + fake.mean())) < 1e-5 and abs(float(g_loss) + fake.mean()) < 1e-5
    assert relerr(realt.grad, np.full(7, -1 / 7)) < BF_TOL and relerr(faket.grad, np.full(5, 1 / 5)) < BF_TOL
    with pytest.raises(NotImplementedError):
        misc.get_loss(realt, faket, 'no-such-loss')
    # the penalty the reference pastes at the call site (misc.py:341-349): value and derivative
    gsrc, gt = bf(rng.normal(size=(6, 4, 4, 3)))
    gt.requires_grad_(True)
    gp = misc.gradient_penalty(gt, 10.0)
    gp.backward()
    gr = torch.tensor(gsrc, requires_grad=True)
    ref = 10. * ((torch.sqrt((gr ** 2).sum(dim=(1, 2, 3)) + 1e-10) - 1.) ** 2).mean()
    ref.backward()
    assert abs(float(gp) - float(ref)) < 1e-3 * float(ref) and relerr(gt.grad, gr.grad.numpy()) < BF_TOL


@pytest.mark.parametrize("n,h,cin,cout,k", [(3, 8, 64, 128, 3), (2, 16, 128, 128, 3), (2, 32, 64, 256, 3), (2, 8, 64, 64, 1), (1, 6, 64, 96, 3)])
def test_conv_fprop_upsampled_residual(K, n, h, cin, cout, k):
    """GANK_RES_UPSAMPLE2X: the shortcut of an 'up' block (gan_cifar_resnet.py:179-182,209) stays at half resolution
    and is added nearest-neighbour upsampled in the epilogue (generic and patch kernels)."""
    rng = np.random.default_rng(n + h + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    w, _ = bf(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin))
    b, bt = f32(rng.normal(size=cout))
    res, rest = bf(rng.normal(size=(n, h // 2, h // 2, cout)))
    wf, _ = K.prep_weights(torch.tensor(w, dtype=torch.float32).cuda(), True, False)
    y = K.conv2d_fprop(xt, wf, bt, (h, h), cout, k, K.RES_UPSAMPLE2X, 1.0, rest)
    ref = R.conv2d_same(x, w, b) + R.upsample_nn2x(res)
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL


@pytest.mark.parametrize("slabs,n", [(False, 16), (True, 16), (True, 128)])
def test_conv_wgrad_batched_equals_separate(K, slabs, n):
    """Four same-shape 3x3 filter gradients (the critic's 8x8x128 blocks) in one launch == four separate launches.  slabs: the
    partial tiles of the pixel splits go to slabs and ONE gank_sum_slabs launch adds them into the (non-zero) targets with the
    scale -- no fp32 atomics, the same result on every run."""
    rng = np.random.default_rng(8)
    h, c = 8, 128
    items, refs = [], []
    for i in range(5):                      # 4 in one launch + 1 through the single-layer path
        x, xt = bf(rng.normal(size=(n, h, h, c)))
        dy, dyt = bf(rng.normal(size=(n, h, h, c)))
        dw = torch.full((3, 3, c, c), float(i), dtype=torch.float32, device="cuda")
        db = torch.zeros(c, dtype=torch.float32, device="cuda") if i % 2 == 0 else None
        items.append((xt, dyt, dw, db))
        _, dw_ref, db_ref = R.conv2d_same_grads(R.relu(x), np.zeros((3, 3, c, c)), dy)
        refs.append((dw_ref, db_ref))
    jobs = [] if slabs else None
    K.conv2d_wgrad_batched(items, (h, h), 3, K.IN_RELU, 1.0, slab_jobs=jobs)
    if slabs:
        assert len(jobs) == 1 and jobs[0][1] == 5                  # every layer has its job; nothing reached the targets yet
        assert all(float((dw - float(i)).abs().max()) == 0.0 for i, (_, _, dw, _) in enumerate(items))
        K.sum_slabs(jobs)
        assert jobs == []
    torch.cuda.synchronize()
    for i, ((_, _, dw, db), (dw_ref, db_ref)) in enumerate(zip(items, refs)):
        assert relerr(dw - float(i), dw_ref) < F32_FROM_BF_TOL
        if db is not None:
            assert relerr(db, db_ref) < F32_FROM_BF_TOL
    if slabs:                # deterministic: a second run from the same targets is bit-identical
        first = [dw.clone() for _, _, dw, _ in items]
        for i, (_, _, dw, _) in enumerate(items):
            dw.fill_(float(i))
        K.conv2d_wgrad_batched(items, (h, h), 3, K.IN_RELU, 1.0, slab_jobs=jobs)
        K.sum_slabs(jobs)
        torch.cuda.synchronize()
        assert all(torch.equal(a, dw) for a, (_, _, dw, _) in zip(first, items))


@pytest.mark.parametrize("n,h,w_,cin,cout,relu", [(3, 8, 16, 64, 128, True), (5, 16, 8, 128, 64, False), (24, 8, 8, 256, 256, True)])
def test_conv_wgrad_filter_row_kernel(K, n, h, w_, cin, cout, relu):
    """Plain 3x3 filter gradients below the all-taps kernel's size run on the filter-row kernel (one block = the three taps of
    one filter row of a 64x64 channel tile; several 8x8 patches per image, non-square grids, pixel splits with atomics)."""
    rng = np.random.default_rng(n + h + w_ + cin)
    x, xt = bf(rng.normal(size=(n, h, w_, cin)))
    dy, dyt = bf(rng.normal(size=(n, h, w_, cout)))
    dw = torch.full((3, 3, cin, cout), 2.0, dtype=torch.float32, device="cuda")
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.prof_enable(True)
    K.prof_reset()
    K.conv2d_wgrad(xt, dyt, dw, (h, w_), 3, K.IN_RELU if relu else 0, 1.0, dbias=db)
    torch.cuda.synchronize()
    ran = [k[0] for k in K.prof_kernels(1)]
    K.prof_enable(False)
    assert any("conv_wgrad_rows_kernel" in k for k in ran), ran
    _, ref, refb = R.conv2d_same_grads(R.relu(x) if relu else x, np.zeros((3, 3, cin, cout)), dy)
    assert relerr(dw - 2.0, ref) < F32_FROM_BF_TOL
    assert relerr(db, refb) < F32_FROM_BF_TOL


@pytest.mark.parametrize("n,hw,c", [(3, 4, 64), (5, 16, 128), (2, 32, 256)])
def test_layer_instance_pixel_norm_ops(K, n, hw, c):
    """layer_norm / instance_norm / pixel_norm of common/ops/normalization.py:62-140 through the reference-shaped ops."""
    from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
    from gan_lib_tensorflow_amd.common.ops import normalization as Nm
    rng = np.random.default_rng(n + hw + c)
    store = set_default_store(ParamStore("cuda", seed=0))
    x, xt0 = bf(rng.normal(size=(n, hw, hw, c)) * 1.5 + 0.4)
    dy, dyt = bf(rng.normal(size=x.shape))
    gamma = (rng.normal(size=c) + 1.0).astype(np.float32)
    beta = rng.normal(size=c).astype(np.float32)

    def run(op, scope, gname, bname, shape):
        xt = xt0.clone().requires_grad_(True)
        with store.variable_scope(scope):
            op(xt)                                            # creates gamma / beta
        with torch.no_grad():
            store.vars[gname].copy_(torch.tensor(gamma).view(shape))
            store.vars[bname].copy_(torch.tensor(beta).view(shape))
        store.vars[gname].grad = store.vars[bname].grad = None
        with store.variable_scope(scope):
            y = op(xt)
        y.backward(dyt)
        torch.cuda.synchronize()
        return y, xt.grad, store.vars[gname].grad, store.vars[bname].grad

    # layer norm
    y, dx, dg, db = run(lambda t: Nm.layer_norm('D.Block.2.N1', [1, 2, 3], t), 'Discriminator', 'Discriminator/D.Block.2.N1/gamma',
                        'Discriminator/D.Block.2.N1/beta', (c,))
    ref, cache = R.layer_norm_forward(x, gamma, beta)
    dx_ref, dg_ref, db_ref = R.layer_norm_backward(dy, gamma, cache)
    assert relerr(y, ref) < BF_TOL and relerr(dx, dx_ref) < 2 * BF_TOL
    assert relerr(dg, dg_ref) < F32_FROM_BF_TOL and relerr(db, db_ref) < F32_FROM_BF_TOL
    # instance norm
    y, dx, dg, db = run(lambda t: Nm.instance_norm(t), 'G.IN', 'G.IN/InstanceNorm/gamma', 'G.IN/InstanceNorm/beta', (1, c))
    ref, cache = R.instance_norm_forward(x, gamma, beta)
    dx_ref, dg_ref, db_ref = R.cond_batchnorm_backward(dy, np.zeros(n, np.int64), gamma.reshape(1, c).astype(np.float64), cache, n)
    assert relerr(y, ref) < BF_TOL and relerr(dx, dx_ref) < 2 * BF_TOL
    assert relerr(dg, dg_ref) < F32_FROM_BF_TOL and relerr(db, db_ref) < F32_FROM_BF_TOL
    # pixel norm
    xt = xt0.clone().requires_grad_(True)
    y = Nm.pixel_norm(xt)
    y.backward(dyt)
    torch.cuda.synchronize()
    assert relerr(y, R.pixel_norm_forward(x)) < BF_TOL and relerr(xt.grad, R.pixel_norm_backward(dy, x)) < BF_TOL


def _rand_conv_cases(seed, count):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        k = int(rng.choice([1, 3]))
        cin = int(rng.choice([3, 8, 64, 96, 128, 256]))
        cout = int(rng.choice([3, 8, 32, 64, 128, 192]))
        h = int(rng.choice([1, 2, 3, 5, 8, 16]))
        w = int(rng.choice([1, 2, 4, 6, 8, 16, 32]))
        n = int(rng.integers(1, 5))
        cases.append((n, h, w, cin, cout, k, bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("n,h,w,cin,cout,k,relu", _rand_conv_cases(2024, 24) + [(2, 8, 16, 64, 128, 3, True), (1, 16, 32, 128, 128, 3, False),
                                                                               (3, 8, 32, 64, 3, 3, False), (2, 4, 16, 3, 128, 3, False)])
def test_conv_random_and_non_square_shapes(K, n, h, w, cin, cout, k, relu):
    """Seeded random geometry sweep, including non-square images (H != W) that no reference layer uses but the ABI
    accepts: fprop (+bias), dgrad and wgrad (+bias gradient) of the same layer against the oracle."""
    rng = np.random.default_rng(n * 131 + h * 17 + w * 5 + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, w, cin)))
    wgt, _ = bf(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin))
    b, bt = f32(rng.normal(size=cout))
    dy, dyt = bf(rng.normal(size=(n, h, w, cout)))
    wt = torch.tensor(wgt, dtype=torch.float32).cuda()
    wf, wd = K.prep_weights(wt, True, True)
    flags = K.IN_RELU if relu else 0
    xin = R.relu(x) if relu else x
    y = K.conv2d_fprop(xt, wf, bt, (h, w), cout, k, flags)
    dx = K.conv2d_dgrad(dyt, wd, (h, w), cin, k)
    dw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device="cuda")
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.conv2d_wgrad(xt, dyt, dw, (h, w), k, flags, 1.0, dbias=db)
    dx_ref, dw_ref, db_ref = R.conv2d_same_grads(xin, wgt, dy)
    torch.cuda.synchronize()
    assert relerr(y, R.conv2d_same(xin, wgt, b)) < BF_TOL
    assert relerr(dx, dx_ref) < BF_TOL                        # gradient w.r.t. the (relu'd) conv operand
    assert relerr(dw, dw_ref) < F32_FROM_BF_TOL
    assert relerr(db, db_ref) < F32_FROM_BF_TOL


# ---- two-group LDS-DMA patch kernel (conv_igemm_pp_kernel): taken only when a full round of 512-thread blocks exists, so
# these cases are sized to >= 224 blocks of 256 pixels x 256 couts.
@pytest.mark.parametrize("n,h,w_,cin,cout,opts", [
    (56, 32, 32, 64, 256, "bias,res"),              # 8 x 32 patches (one-row pixel tiles)
    (28, 32, 64, 64, 256, "relu,bias"),             # two patches per image row
    (56, 32, 32, 128, 256, "tanh,scale"),           # 4 chunks of 32 channels
    (224, 16, 16, 64, 256, "relu,res_up"),          # 16 x 16 patches (two-row pixel tiles), residual stored at half size
    (112, 16, 16, 64, 512, ""),                     # two cout tiles
])
def test_conv3x3_two_group_kernel_fprop(K, n, h, w_, cin, cout, opts):
    rng = np.random.default_rng(n + h + cin + cout)
    x, xt = bf(rng.normal(size=(n, h, w_, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    wf, _ = K.prep_weights(wt, True, False)
    b, bt = f32(rng.normal(size=cout)) if "bias" in opts else (0.0, None)
    flags = (K.IN_RELU if "relu" in opts else 0) | (K.OUT_TANH if "tanh" in opts else 0)
    scale = 0.5 if "scale" in opts else 1.0
    res, rest = None, None
    if "res_up" in opts:
        res, rest = bf(rng.normal(size=(n, h // 2, w_ // 2, cout)))
        flags |= K.RES_UPSAMPLE2X
    elif "res" in opts:
        res, rest = bf(rng.normal(size=(n, h, w_, cout)))
    K.prof_reset()
    K.prof_enable(True)
    y = K.conv2d_fprop(xt, wf, bt, (h, w_), cout, 3, flags, scale, rest)
    K.prof_enable(False)
    K.prof_collect(0)
    ran = [k[0] for k in K.prof_kernels(0)]
    K.prof_reset()
    assert any("conv_igemm_pp_kernel" in k for k in ran), ran      # the case really exercises the two-group kernel
    ref = R.conv2d_same(R.relu(x) if "relu" in opts else x, w, np.zeros(cout)) * scale + b
    if res is not None:
        ref = ref + (R.upsample_nn2x(res) if "res_up" in opts else res)
    if "tanh" in opts:
        ref = np.tanh(ref)
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL


@pytest.mark.parametrize("n,h,cin,cout,mask", [(56, 32, 256, 64, True), (224, 16, 256, 64, False)])
def test_conv3x3_two_group_kernel_dgrad(K, n, h, cin, cout, mask):
    """Input gradient of a 3x3 conv = the same kernel over dy with the flipped filter; relu mask in its epilogue."""
    rng = np.random.default_rng(n + h + cin)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cout))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    _, wd = K.prep_weights(wt, False, True)
    dy, dyt = bf(rng.normal(size=(n, h, h, cout)))
    dx = K.conv2d_dgrad(dyt, wd, (h, h), cin, 3, 0, 1.0, None, xt if mask else None)
    ref, _, _ = R.conv2d_same_grads(x, w, dy)
    if mask:
        ref = ref * (x > 0)
    torch.cuda.synchronize()
    assert relerr(dx, ref) < BF_TOL


@pytest.mark.parametrize("n,hl,wl,cin,cout", [(56, 16, 16, 64, 256), (28, 8, 32, 64, 256), (28, 16, 16, 128, 512)])
def test_two_group_kernel_phase_form(K, n, hl, wl, cin, cout):
    """The four output phases of the stride-2 transposed convs on the two-group kernel: UpsampleConv 3x3 fprop over a
    low-res grid, and the ConvMeanPool 3x3 input gradient (relu mask in the epilogue)."""
    rng = np.random.default_rng(n + hl + cin)
    x, xt = bf(rng.normal(size=(n, hl, wl, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout))
    res, rest = bf(rng.normal(size=(n, 2 * hl, 2 * wl, cout)))
    wph, _ = K.upconv3x3_prep(torch.tensor(w, dtype=torch.float32).cuda())
    y = K.upconv3x3_fprop(xt, wph, bt, cout, 0, rest)
    torch.cuda.synchronize()
    assert relerr(y, R.conv2d_same(R.upsample_nn2x(x), w, b) + res) < BF_TOL
    # ConvMeanPool with cin2 = cout (a multiple of 256) input channels: dy [n,hl,wl,cin] -> dx [n,2hl,2wl,cout]
    x2, x2t = bf(rng.normal(size=(n, 2 * hl, 2 * wl, cout)))
    w2, _ = bf(rng.normal(size=(3, 3, cout, cin)) / np.sqrt(9 * cin))
    _, wphd = K.convpool3x3_prep(torch.tensor(w2, dtype=torch.float32).cuda())
    dy, dyt = bf(rng.normal(size=(n, hl, wl, cin)))
    dx = K.convpool3x3_dgrad(dyt, wphd, cout, x2t)
    dx_ref, _, _ = R.conv2d_same_grads(R.relu(x2), w2, R.meanpool2x2_grad(dy))
    torch.cuda.synchronize()
    assert relerr(dx, dx_ref * (x2 > 0)) < BF_TOL


@pytest.mark.parametrize("form,n,h,w_,cin,cout", [("plain", 320, 32, 32, 64, 512), ("plain", 320, 16, 16, 256, 256), ("phase", 128, 16, 16, 256, 256)])
def test_two_group_kernel_race_screen(K, form, n, h, w_, cin, cout):
    """The two-group kernel has no atomics: every repetition over the same operands must be BIT-identical to the first.
    A staged LDS buffer read before its DMA landed shows up as a differing tile (a prologue wait that left K-step 1 in
    flight passed every parity test above and failed this screen in 1-30 % of the launches).  A second stream streams
    HBM beside the kernel to perturb DMA latency."""
    g = torch.Generator(device="cpu").manual_seed(n + cin)
    x = torch.randn((n, h, w_, cin), generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / (9 * cin) ** 0.5).cuda()
    if form == "plain":
        wf, _ = K.prep_weights(w, True, False)
        run = lambda: K.conv2d_fprop(x, wf, None, (h, w_), cout, 3)     # noqa: E731
    else:
        wph, _ = K.upconv3x3_prep(w)
        run = lambda: K.upconv3x3_fprop(x, wph, None, cout)              # noqa: E731
    side, junk = torch.cuda.Stream(), torch.randn(32 << 20, device="cuda")
    ref = run().clone()
    torch.cuda.synchronize()
    differing = 0
    for rep in range(120):
        if rep % 3 == 0:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        differing += 0 if torch.equal(run(), ref) else 1
    torch.cuda.synchronize()
    assert differing == 0


@pytest.mark.parametrize("n,nb,pool", [(3, 2, True), (5, 1, False), (2, 2, False), (128, 2, True), (1, 1, True)])
def test_res8_chain_fused_residual_blocks(K, n, nb, pool):
    """conv_resident.hip: nb identity-shortcut residual blocks on 8x8x128 images (+ relu + spatial mean) in one launch each
    way, against torch-CPU float64 autograd of gan_cifar_resnet.py:176-209,299-301 on the same bf16-rounded operands.  The
    HIP path stores h1 and every block output as bf16 and takes its relu masks from those tensors; the oracle rounds the
    same tensors to bf16 with a straight-through estimator, so both sides differentiate through the same masks (against
    unrounded float64 intermediates ~0.1 % of the masks flip and the gradients differ by 3e-2 in L2 -- a property of bf16
    storage, not of the kernel).  Outputs and input gradients <= 1.5e-2 of the maximum; filter / bias gradients <= 1e-2."""
    from oracle import ref_torch as T
    from gan_lib_tensorflow_amd import functional as Fn
    rng = np.random.default_rng(800 + n + 10 * nb + pool)
    x, xt = bf(rng.normal(size=(n, 8, 8, 128)))
    params_t, params_r = [], []
    for b in range(nb):
        blk_t, blk_r = [], []
        for j in range(2):
            w, _ = bf(rng.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4)
            bias, biast = f32(rng.normal(size=128) * 0.1)
            wt = torch.tensor(w, dtype=torch.float32).cuda().requires_grad_(True)
            biast.requires_grad_(True)
            blk_t += [wt, biast]
            blk_r += [torch.tensor(w, requires_grad=True), torch.tensor(bias, requires_grad=True)]
        params_t.append(tuple(blk_t))
        params_r.append(blk_r)
    K.prep_weights_batched([p[i] for p in params_t for i in (0, 2)], want_d=True, kinds=[4] * (2 * nb))
    xt.requires_grad_(True)
    out = Fn.res_chain8(xt, params_t, pool=pool)
    g, gt = bf(rng.normal(size=tuple(out.shape)))
    out.backward(gt)
    torch.cuda.synchronize()
    xr = torch.tensor(x, requires_grad=True)
    cur = xr
    def rb(t):          # bf16 storage, identity gradient
        return t + (t.detach().to(torch.bfloat16).to(torch.float64) - t.detach())
    for w1, b1, w2, b2 in params_r:
        h = rb(T.conv2d_same(torch.relu(cur), w1, b1))
        cur = rb(cur + T.conv2d_same(torch.relu(h), w2, b2))
    ref = torch.relu(cur).mean(dim=(1, 2)) if pool else cur
    ref.backward(torch.tensor(g))
    assert relerr(out, ref.detach().numpy()) < 1.5e-2
    # a mask still flips where fp32-accumulated and float64 values round to different sides of zero (a few elements in a
    # million): bound the input gradient in L2 and the fraction of elements off by more than the max-norm tolerance
    # ONE flipped mask (an h1 / block-output element whose value sits within rounding of zero: which side it lands on depends on
    # the summation order of the fp32 accumulation, e.g. on how the reduction is split over waves) moves up to 9 pixels x 128
    # channels of the input gradient, 9 x 128 elements of two filter gradients and (slightly) the bias gradients in front of it.  The small cases
    # (16 K gradient elements) are therefore allowed two flips' worth of bounded outliers; the remaining elements keep an L2 bound of
    # 1e-2 (2e-2 where a flip's sub-threshold tail is in them).
    def close_up_to_mask_flips(got, ref, tol, per_flip, what):
        gd = (got.double().cpu() - ref).abs()
        top = float(ref.abs().max())
        outl = gd > tol * top
        allowed = max(5e-4 * gd.numel(), 2 * min(per_flip, gd.numel()))
        rest = float(gd[~outl].norm() / ref.norm()) if bool((~outl).any()) else 0.0
        assert int(outl.sum()) <= allowed and float(gd.max()) < 0.5 * top and rest < (2e-2 if bool(outl.any()) else 1e-2), (what, int(outl.sum()), allowed, float(gd.max()) / top, rest)
    close_up_to_mask_flips(xt.grad, xr.grad, 1.5e-2, 9 * 128, "dx")        # measured 5.5e-3 L2 / 1.6e-4 outliers on 1 M elements through 4 masks
    for b in range(nb):
        for i in range(4):
            close_up_to_mask_flips(params_t[b][i].grad, params_r[b][i].grad, 1e-2, 9 * 128 if i % 2 == 0 else 128, (b, i))      # (a flip in a LATER block reaches every channel of an earlier bias gradient)
    # inference form (nothing requires a gradient): same output, nothing kept
    with torch.no_grad():
        out2 = Fn.res_chain8(xt, params_t, pool=pool)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize("n,n_real,mode,scale", [(128, 64, 0, 1.0), (6, 2, 0, 1.0), (7, 0, 1, 1.0), (128, 0, 1, 1.0), (16, 8, 0, 1024.0)])
def test_res8_chain_with_the_critic_head_inside(K, n, n_real, mode, scale):
    """gank_res8_chain_fwd_head / _bwd_head: D.Output + hinge loss (gan_cifar_resnet.py:303-304, :379-381 / :492) computed inside
    the fused chain's two launches.  Reference = the SAME chain followed by the stand-alone head launch
    (gank_critic_head_hinge_scaled, itself tested against linear + hinge and the oracle): logits, the chain's input gradient
    and every filter / bias gradient behind it are BIT-identical (the in-chain head repeats the head kernel's arithmetic and
    rounding); the loss and the head's own weight / bias gradient sum the same terms in another order (fp32: <= 1e-6 relative).
    Also: the forward-only form returns the loss at once; the loss against the float64 oracle."""
    from oracle import ref_ops as R
    from gan_lib_tensorflow_amd import functional as Fn
    rng = np.random.default_rng(4100 + n + mode)
    x, xt0 = bf(rng.normal(size=(n, 8, 8, 128)))
    hw = (rng.normal(size=(128, 1)) * 0.6).astype(np.float32)
    hb = np.asarray([0.13], np.float32)

    def build():
        r2 = np.random.default_rng(77)
        params = []
        for b in range(2):
            blk = []
            for j in range(2):
                w, _ = bf(r2.normal(size=(3, 3, 128, 128)) / np.sqrt(9 * 128) * 1.4)
                blk += [torch.tensor(w, dtype=torch.float32).cuda().requires_grad_(True), torch.tensor(r2.normal(size=128).astype(np.float32) * 0.1).cuda().requires_grad_(True)]
            params.append(tuple(blk))
        K.prep_weights_batched([p[i] for p in params for i in (0, 2)], want_d=True, kinds=[4] * 4)
        return params, torch.tensor(hw).cuda().requires_grad_(True), torch.tensor(hb).cuda().requires_grad_(True), xt0.clone().requires_grad_(True)

    res = {}
    for form in ("separate", "fused"):
        params, W, b, xt = build()
        out = torch.zeros(1, dtype=torch.float32, device="cuda")
        spec = Fn.HingeHeadSpec(mode, n_real, out=out, loss_scale=scale)
        if form == "separate":
            f = Fn.res_chain8(xt, params, pool=True)
            loss = spec(f, W, b)
        else:
            loss = Fn.res_chain8(xt, params, pool=True, head=(spec, W, b))
        logits = loss.logits
        loss.backward(gradient=Fn.grad_seed(loss, scale))
        torch.cuda.synchronize()
        res[form] = dict(loss=float(out), logits=logits.clone(), dx=xt.grad.clone(), W=W.grad.clone(), b=b.grad.clone(),
                         params=[p.grad.clone() for blk in params for p in blk])
    a, c = res["separate"], res["fused"]
    assert torch.equal(a["logits"], c["logits"]) and torch.equal(a["dx"], c["dx"])
    for pa, pc in zip(a["params"], c["params"]):
        # the filter gradients of the chain's 8x8 layers accumulate with fp32 atomics: same operands, arrival order differs
        assert float((pa - pc).abs().max()) <= 2e-5 * float(pa.abs().max()) + 1e-12
    assert abs(a["loss"] - c["loss"]) <= 1e-6 * max(1.0, abs(a["loss"]))
    assert float((a["W"] - c["W"]).abs().max()) <= 1e-6 * float(a["W"].abs().max()) + 1e-12, float((a["W"] - c["W"]).abs().max())
    assert float((a["b"] - c["b"]).abs().max()) <= 1e-6 * max(1.0, float(a["b"].abs().max()))
    assert float(c["dx"].abs().max()) > 0
    # the loss against the oracle on the fused logits
    lg = c["logits"].double().cpu().numpy()
    want = float(R.hinge_d_loss(lg, n_real)[0]) if mode == 0 else float(R.hinge_g_loss(lg)[0])
    assert abs(c["loss"] - want) < 1e-5 * max(1.0, abs(want)), (c["loss"], want)
    # forward only (no gradient anywhere): the loss is there without a backward launch
    params, W, b, xt = build()
    out = torch.full((1,), -7.0, dtype=torch.float32, device="cuda")
    with torch.no_grad():
        loss = Fn.res_chain8(xt, params, pool=True, head=(Fn.HingeHeadSpec(mode, n_real, out=out, loss_scale=scale), W, b))
    torch.cuda.synchronize()
    assert torch.equal(loss.logits, c["logits"]) and abs(float(out) - c["loss"]) <= 1e-6 * max(1.0, abs(c["loss"]))


@pytest.mark.parametrize("n,hp,wp,cin,relu", [(3, 16, 16, 128, True), (2, 8, 8, 256, True), (2, 8, 16, 128, False), (5, 8, 8, 128, False),
                                              (2, 16, 32, 256, True)])
def test_convpool3x3_resident_kernels(K, n, hp, wp, cin, relu):
    """conv_resident.hip: ConvMeanPool 3x3 forward (4x4 stride-2 conv out of parity-plane LDS images) and input gradient
    (4 phases out of one resident dy patch), prep kind 5 -- against the un-reduced oracle composition
    mean_pool(conv3x3(relu?(x)) + b) + residual and its gradient, and bit-for-bit against nothing: the implicit-GEMM form
    (gank_convpool3x3_*) sums the same bf16 products in a different order."""
    cout = 128
    rng = np.random.default_rng(n * 11 + hp + wp + cin)
    x, xt = bf(rng.normal(size=(n, 2 * hp, 2 * wp, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout))
    res, rest = bf(rng.normal(size=(n, hp, wp, cout)))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    assert K.cpool_res_ok(n, hp, wp, cin, cout)
    (rf, rd), = K.prep_weights_batched([wt], want_d=True, kinds=[5])
    xin = R.relu(x) if relu else x
    y = K.cpool_res_fprop(xt, rf, bt, cout, K.IN_RELU if relu else 0, rest)
    ref = R.meanpool2x2(R.conv2d_same(xin, w, b)) + res
    torch.cuda.synchronize()
    assert relerr(y, ref) < BF_TOL
    y2 = K.cpool_res_fprop(xt, rf, None, cout, K.IN_RELU if relu else 0, None)          # no bias, no residual
    assert relerr(y2, R.meanpool2x2(R.conv2d_same(xin, w))) < BF_TOL
    dy, dyt = bf(rng.normal(size=(n, hp, wp, cout)))
    dx_ref, _, _ = R.conv2d_same_grads(xin, w, R.meanpool2x2_grad(dy))
    if relu:
        dx_ref = dx_ref * (x > 0)
    dx = K.cpool_res_dgrad(dyt, rd, cin, xt if relu else None)
    torch.cuda.synchronize()
    assert relerr(dx, dx_ref) < BF_TOL
    # the resident and the implicit-GEMM forms agree to bf16 rounding of the same operands
    wp4, wphd = K.convpool3x3_prep(wt)
    y_ig = K.convpool3x3_fprop(xt, wp4, bt, cout, K.IN_RELU if relu else 0, rest)
    dx_ig = K.convpool3x3_dgrad(dyt, wphd, cin, xt if relu else None)
    assert relerr(y, y_ig.double().cpu().numpy()) < 1e-2 and relerr(dx, dx_ig.double().cpu().numpy()) < 1e-2


@pytest.mark.parametrize("form,n,h,groups", [("plain", 64, 32, 4), ("up", 64, 16, 2), ("plain", 8, 8, 2), ("up", 8, 4, 2), ("plain", 128, 8, 2),
                                             ("up", 128, 8, 2), ("plain", 320, 8, 10), ("up", 320, 4, 10)])
def test_conv_epilogue_statistics_feed_cond_batchnorm(K, form, n, h, groups):
    """gank_conv2d_fprop_stats / gank_upconv3x3_fprop_stats + gank_cbn_fwd_from_sums: the batch-norm statistics of a conv
    output from the conv's own epilogue (the two-group kernel at the large shapes, the generic implicit-GEMM kernel at the
    small ones and in its phase form) equal the moments of the stored tensor, and the conditional batch norm fed by them
    equals the three-pass one (mean / invstd <= 1e-3 relative: the epilogue sums the fp32 values before their bf16 rounding)."""
    rng = np.random.default_rng(n + h + groups)
    cin = cout = 256
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout) * 3.0)                        # a large bias: the shift matters
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    if form == "plain":
        wf, _ = K.prep_weights(wt, True, False)
        res = torch.randn(n, h, h, cout, device="cuda").to(torch.bfloat16)
        y, cs = K.conv2d_fprop(xt, wf, bt, (h, h), cout, 3, 0, 1.0, res, stats_groups=groups)
        y0 = K.conv2d_fprop(xt, wf, bt, (h, h), cout, 3, 0, 1.0, res)
    else:
        wph, _ = K.upconv3x3_prep(wt)
        y, cs = K.upconv3x3_fprop(xt, wph, bt, cout, 0, None, stats_groups=groups)
        y0 = K.upconv3x3_fprop(xt, wph, bt, cout)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)                                       # the statistics do not touch the output
    assert cs is not None and cs.groups == groups
    yd = y.double().cpu().numpy().reshape(groups, -1, cout)
    M = yd.shape[1]
    tot = cs.sums.double().sum(dim=1).cpu().numpy()               # add the partial copies
    mean = tot[:, 0] / M + b
    var = tot[:, 1] / M - (tot[:, 0] / M) ** 2
    # (few samples per tower: the bf16 rounding of the stored tensor no longer averages out of its variance)
    assert np.abs(mean - yd.mean(1)).max() < 1e-3 * np.abs(yd).max() and np.abs(var / yd.var(1) - 1).max() < (2e-3 if M >= 16384 else 8e-3)
    labels = torch.tensor(rng.integers(0, 10, n), dtype=torch.int32).cuda()
    gamma = torch.tensor(rng.normal(size=(10, cout)) * 0.2 + 1, dtype=torch.float32).cuda()
    beta = torch.tensor(rng.normal(size=(10, cout)) * 0.2, dtype=torch.float32).cuda()
    z1, s1 = K.cbn_fwd_from_sums(y, labels, gamma, beta, cs, relu=True)
    z0, s0 = K.cbn_fwd(y, labels, gamma, beta, groups, True)
    torch.cuda.synchronize()
    assert relerr(s1[:, 0], s0[:, 0].double().cpu().numpy()) < 1e-3 and relerr(s1[:, 1], s0[:, 1].double().cpu().numpy()) < (2e-3 if M >= 16384 else 4e-3)
    assert relerr(z1, z0.double().cpu().numpy()) < BF_TOL


@pytest.mark.parametrize("n,cin,cout,relu,masked,res", [(2, 256, 256, True, False, False), (3, 256, 256, False, True, True), (1, 128, 128, False, False, False),
                                                       (2, 128, 384, True, True, False), (128, 256, 256, True, False, False), (128, 256, 256, False, True, False)])
def test_img16_conv3x3_resident_image_kernel(K, n, cin, cout, relu, masked, res):
    """gank_img16_conv3x3 (one 16x16 image x 128 output channels per workgroup, chunk images resident in LDS, fragment-major
    weights): forward with the pre-activation relu and bias, and -- through the dgrad operand with the channel roles swapped --
    the input gradient with its relu mask and a gradient fan-in, against the float64 oracle (small cases) and against the
    implicit-GEMM kernels on the same operands (every case, incl. the critic's n = 128)."""
    rng = np.random.default_rng(1600 + n + cin + cout + relu)
    x, xt = bf(rng.normal(size=(n, 16, 16, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout) * 0.5)
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    (rf, rd), = K.prep_weights_batched([wt], want_d=True, kinds=[4])
    wf, wd = K.prep_weights(wt, True, True)
    # forward: y = conv(relu?(x)) + b
    y = K.img16_conv3x3(xt, rf, bt, cout, K.IN_RELU if relu else 0)
    y_ig = K.conv2d_fprop(xt, wf, bt, (16, 16), cout, 3, K.IN_RELU if relu else 0)
    torch.cuda.synchronize()
    assert relerr(y, y_ig.double().cpu().numpy()) < 5e-3
    if n <= 3:
        assert relerr(y, R.conv2d_same(R.relu(x) if relu else x, w, b)) < BF_TOL
    # input gradient: dx = conv(dy, flip(w)^T) [* (mask > 0)] [+ fan-in]
    dy, dyt = bf(rng.normal(size=(n, 16, 16, cout)))
    mk = mkt = rs = rst = None
    if masked:
        mk, mkt = bf(rng.normal(size=(n, 16, 16, cin)))
    if res:
        rs, rst = bf(rng.normal(size=(n, 16, 16, cin)))
    dx = K.img16_conv3x3(dyt, rd, None, cin, 0, relu_ref=mkt, residual=rst)
    dx_ig = K.conv2d_dgrad(dyt, wd, (16, 16), cin, 3, 0, 1.0, rst, mkt)
    torch.cuda.synchronize()
    assert relerr(dx, dx_ig.double().cpu().numpy()) < 5e-3
    if n <= 3:
        rdx, _, _ = R.conv2d_same_grads(x, w, dy)
        if masked:
            rdx = rdx * (mk > 0)
        if res:
            rdx = rdx + rs
        assert relerr(dx, rdx) < BF_TOL


@pytest.mark.parametrize("n,groups,resmode", [(4, 2, "half"), (6, 3, None), (128, 2, "half"), (320, 10, "half")])
def test_img16_conv3x3_statistics_and_half_resolution_residual(K, n, groups, resmode):
    """gank_img16_conv3x3_stats as the generator's G.Block.2.Conv2 uses it: bias, the 'up' block's shortcut added from HALF
    resolution (GANK_RES_UPSAMPLE2X), and the batch-norm statistics of the result accumulated by the epilogue -- against the oracle
    (small cases), the implicit-GEMM kernel on the same operands, and the moments of the stored tensor."""
    rng = np.random.default_rng(1700 + n)
    x, xt = bf(rng.normal(size=(n, 16, 16, 256)))
    w, _ = bf(rng.normal(size=(3, 3, 256, 256)) / np.sqrt(9 * 256))
    b, bt = f32(rng.normal(size=256) * 2.0)
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    (rf, rd), = K.prep_weights_batched([wt], want_d=True, kinds=[4])
    res = rest = None
    if resmode == "half":
        res, rest = bf(rng.normal(size=(n, 8, 8, 256)))
    flags = K.RES_UPSAMPLE2X if resmode == "half" else 0
    y, cs = K.img16_conv3x3(xt, rf, bt, 256, flags, None, rest, stats_groups=groups)
    torch.cuda.synchronize()
    if n <= 6:
        ref = R.conv2d_same(x, w, b)
        if res is not None:
            ref = ref + R.upsample_nn2x(res)
        assert relerr(y, ref) < BF_TOL
    wf, _ = K.prep_weights(wt, True, False)
    y_ig = K.conv2d_fprop(xt, wf, bt, (16, 16), 256, 3, flags, 1.0, rest)
    torch.cuda.synchronize()
    assert relerr(y, y_ig.double().cpu().numpy()) < 5e-3
    yd = y.double().cpu().numpy().reshape(groups, -1, 256)
    M = yd.shape[1]
    tot = cs.sums.double().sum(dim=1).cpu().numpy()
    mean = tot[:, 0] / M + b
    var = tot[:, 1] / M - (tot[:, 0] / M) ** 2
    assert np.abs(mean - yd.mean(1)).max() < 1e-3 * np.abs(yd).max() and np.abs(var / yd.var(1) - 1).max() < 8e-3
    y2, _ = K.img16_conv3x3(xt, rf, bt, 256, flags, None, rest, stats_groups=groups)      # the statistics do not touch the output
    assert torch.equal(y, y2)


@pytest.mark.parametrize("n,cin,cout,up,resmode,groups", [(3, 256, 256, False, "full", 0), (8, 256, 256, True, "half", 2), (4, 128, 128, False, None, 2),
                                                          (6, 128, 256, True, None, 0), (128, 256, 256, False, "half", 2), (320, 256, 256, True, None, 10)])
def test_res8_conv3x3_resident_generator_layers(K, n, cin, cout, up, resmode, groups):
    """gank_res8_conv3x3 (one LDS-resident 8x8 image per workgroup, fragment-major weights): fprop with the loader's NN-upsample,
    bias, full- and half-resolution residual against the oracle and against the implicit-GEMM kernels on the same operands;
    the epilogue's batch-norm statistics against the moments of the stored tensor; the input gradient through the dgrad operand,
    with the 2x2 sums of the upsample's gradient."""
    rng = np.random.default_rng(n + cin + cout + up)
    hin = 4 if up else 8
    x, xt = bf(rng.normal(size=(n, hin, hin, cin)))
    w, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, bt = f32(rng.normal(size=cout) * 2.0)
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    (rf, rd), = K.prep_weights_batched([wt], want_d=True, kinds=[4])
    res = rest = None
    if resmode is not None:
        rh = 8 if resmode == "full" else 4
        res, rest = bf(rng.normal(size=(n, rh, rh, cout)))
    flags = (K.IN_UPSAMPLE2X if up else 0) | (K.RES_UPSAMPLE2X if resmode == "half" else 0)
    if groups:
        y, cs = K.res8_conv3x3(xt, rf, bt, cout, flags, rest, stats_groups=groups)
    else:
        y = K.res8_conv3x3(xt, rf, bt, cout, flags, rest)
    torch.cuda.synchronize()
    small = n <= 8
    if small:
        xin = R.upsample_nn2x(x) if up else x
        ref = R.conv2d_same(xin, w, b)
        if res is not None:
            ref = ref + (R.upsample_nn2x(res) if resmode == "half" else res)
        assert relerr(y, ref) < BF_TOL
    # the implicit-GEMM kernels on the same operands
    wf, wd = K.prep_weights(wt, True, True)
    y_ig = K.conv2d_fprop(xt, wf, bt, (8, 8), cout, 3, flags, 1.0, rest)
    torch.cuda.synchronize()
    assert relerr(y, y_ig.double().cpu().numpy()) < 5e-3
    if groups:
        yd = y.double().cpu().numpy().reshape(groups, -1, cout)
        M = yd.shape[1]
        tot = cs.sums.double().sum(dim=1).cpu().numpy()
        mean = tot[:, 0] / M + b
        var = tot[:, 1] / M - (tot[:, 0] / M) ** 2
        assert np.abs(mean - yd.mean(1)).max() < 1e-3 * np.abs(yd).max() and np.abs(var / yd.var(1) - 1).max() < 8e-3
        y2, _ = K.res8_conv3x3(xt, rf, bt, cout, flags, rest, stats_groups=groups)      # the statistics do not touch the output
        assert torch.equal(y, y2)
    # input gradient
    dy, dyt = bf(rng.normal(size=(n, 8, 8, cout)))
    dx = K.res8_conv3x3(dyt, rd, None, cin, K.OUT_POOLSUM2X if up else 0)
    torch.cuda.synchronize()
    assert tuple(dx.shape) == (n, hin, hin, cin)
    dx_ig = K.conv2d_dgrad(dyt, wd, (8, 8), cin, 3, 0, 1.0)
    if up:
        dx_ig = K.pool2x2(dx_ig, 1.0)
    torch.cuda.synchronize()
    assert relerr(dx, dx_ig.double().cpu().numpy()) < (BF_TOL if up else 5e-3)     # (pooled: the reference side rounds twice)
    if small:
        dref, _, _ = R.conv2d_same_grads(R.upsample_nn2x(x) if up else x, w, dy)
        if up:
            dref = R.upsample_nn2x_grad(dref)
        assert relerr(dx, dref) < BF_TOL


def test_copy_gather(K):
    """gank_copy_bytes_gather: equal-sized device buffers into consecutive slots of one buffer, one launch (aligned and odd sizes)"""
    for shape, dt in (((64, 3072), torch.uint8), ((64,), torch.int32), ((7, 3), torch.uint8)):
        srcs = [(torch.arange(int(np.prod(shape)), device="cuda") * (i + 3) % 251).to(dt).reshape(shape) for i in range(5)]
        dst = torch.zeros((5,) + shape, dtype=dt, device="cuda")
        K.copy_gather_(dst, srcs)
        torch.cuda.synchronize()
        assert all(torch.equal(dst[i], srcs[i]) for i in range(5))
    # two gathers with different row sizes in one launch (gank_copy_bytes_gather2: the image batches and the label vectors of an iteration)
    for sa, sb in (((64, 3072), (64,)), ((7, 3), (5,))):
        a_src = [(torch.arange(int(np.prod(sa)), device="cuda") * (i + 3) % 251).to(torch.uint8).reshape(sa) for i in range(5)]
        b_src = [(torch.arange(int(np.prod(sb)), device="cuda") * (i + 7) % 11).to(torch.int32).reshape(sb) for i in range(3)]
        da, db = torch.zeros((5,) + sa, dtype=torch.uint8, device="cuda"), torch.full((3,) + sb, -1, dtype=torch.int32, device="cuda")
        K.copy_gather2_(da, a_src, db, b_src)
        torch.cuda.synchronize()
        assert all(torch.equal(da[i], a_src[i]) for i in range(5)) and all(torch.equal(db[i], b_src[i]) for i in range(3))


def test_fork_pool_equals_fork_then_meanpool(K):
    """functional.fork_pool: (x, mean_pool2x2(x)) with ONE unpool-add launch backward == the separate fork + meanpool2x2
    (forward bit-identical; the backward sum has one rounding to bf16 instead of two: compared against float64)"""
    from gan_lib_tensorflow_amd import functional as Fn
    rng = np.random.default_rng(5)
    x, xt = bf(rng.normal(size=(4, 8, 8, 16)))
    ga, gat = bf(rng.normal(size=(4, 8, 8, 16)))
    gp, gpt = bf(rng.normal(size=(4, 4, 4, 16)))
    x1 = xt.clone().requires_grad_(True)
    a1, p1 = Fn.fork_pool(x1)
    torch.autograd.backward([a1, p1], [gat, gpt])
    x2 = xt.clone().requires_grad_(True)
    a2, b2 = Fn.fork(x2)
    p2 = Fn.meanpool2x2(b2)
    torch.autograd.backward([a2, p2], [gat, gpt])
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(p1, p2)
    ref = ga + 0.25 * np.repeat(np.repeat(gp, 2, axis=1), 2, axis=2)
    assert relerr(x1.grad, ref) < BF_TOL and relerr(x2.grad, ref) < BF_TOL
    # only one branch carries a gradient
    x3 = xt.clone().requires_grad_(True)
    a3, p3 = Fn.fork_pool(x3)
    p3.backward(gpt)
    torch.cuda.synchronize()
    assert relerr(x3.grad, 0.25 * np.repeat(np.repeat(gp, 2, axis=1), 2, axis=2)) < BF_TOL


# ---------------------------------------------------------------------------------------------- round 3
def test_spectral_norm_two_launch_form_shapes_and_replays(K):
    """The two-launch forward (row dots + partial column sums in one pass, norms by the last chunk block of each weight) and
    the two-launch backward on every shape class: register-tile form (C a power of two in [4, 256], ragged last chunk),
    generic form (C = 1, 33, 512), one-chunk weights (no ticket), K = 9216 (144 chunks).  Replayed with the weights changed
    in place between calls: a finalizer that read a stale partial sum of the previous call (a missing acquire, a ticket left
    non-zero) shows up as an error against the oracle of the NEW weights."""
    rng = np.random.default_rng(50)
    shapes = [(3, 128), (27, 128), (1152, 128), (300, 128), (2304, 256), (128, 1), (70, 33), (100, 64), (65, 4), (9216, 256), (130, 512), (64, 8), (200, 16)]
    Ws, us, Gs = [], [], []
    for kk, c in shapes:
        Ws.append(f32(rng.normal(size=(kk, c)) * 0.05))
        us.append(f32(rng.normal(size=(1, c))))
        Gs.append(f32(rng.normal(size=(kk, c))))
    batch = K.SnBatch([w[1] for w in Ws], [u[1] for u in us])
    for rep in range(4):
        if rep:
            for w, wt in Ws:
                wt.mul_(1.0 + 0.37 * rep)
                w *= np.float32(1.0 + 0.37 * rep)
            wnow = [(wt.double().cpu().numpy(), wt) for _, wt in Ws]
        else:
            wnow = Ws
        Wbars = batch.forward()
        dWs = [torch.zeros_like(w[1]) for w in Ws]
        batch.backward([g[1] for g in Gs], dWs)
        torch.cuda.synchronize()
        for i, (w, u, g) in enumerate(zip(wnow, us, Gs)):
            Wb, u1, sigma, v = R.sn_forward(w[0], u[0])
            assert relerr(Wbars[i], Wb) < 1e-5, (rep, shapes[i])
            assert relerr(batch.u_out_views()[i], u1.ravel()) < 1e-5, (rep, shapes[i])
            assert abs(float(batch.sigma(i)) - sigma) / sigma < 1e-5, (rep, shapes[i])
            assert relerr(dWs[i], R.sn_backward(w[0], u[0], g[0])) < 2e-4, (rep, shapes[i])


def test_spectral_norm_fused_operand_copies_and_label_table_are_bit_identical(K):
    """gank_sn_power_iter_fwd_prep: the MFMA operand copies built from W / sigma inside the second spectral-norm launch are
    the bytes gank_conv2d_prep_weights_batched builds from the stored W_bar (every operand kind the critic uses), and the
    per-label table is embedding_fwd + linear_fwd on the ten labels (bit for bit from W_bar; one rounding apart from W, sigma)."""
    rng = np.random.default_rng(51)
    specs = [((1, 1, 3, 128), 0), ((3, 3, 3, 128), 0), ((3, 3, 128, 128), 5), ((300, 128), None), ((1, 1, 256, 128), 0), ((3, 3, 256, 256), 0),
             ((3, 3, 256, 128), 5), ((3, 3, 128, 128), 4), ((3, 3, 128, 128), 4), ((128, 1), None), ((3, 3, 64, 64), 2), ((3, 3, 64, 128), 1)]
    Ws = [f32(rng.normal(size=s) * 0.05)[1] for s, _ in specs]
    us = [f32(rng.normal(size=(1, s[-1])))[1] for s, _ in specs]
    kinds = [k for _, k in specs]
    table = f32(rng.uniform(-0.08, 0.08, size=(10, 300)))[1]
    bias = f32(rng.normal(size=128) * 0.1)[1]
    ref = K.SnBatch(Ws, us)
    Wb_ref = ref.forward()
    K.prep_weights_batched(Wb_ref, want_d=True, kinds=kinds)
    emb = K.embedding_fwd(table, torch.arange(10, dtype=torch.int32, device="cuda"))
    T_ref = K.linear_fwd(emb, Wb_ref[3], bias)
    fused = K.SnBatch(Ws, us)
    fused.prep, fused.label = (kinds, True), (table, 3, bias)
    Wb = fused.forward()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(Wb, Wb_ref)):
        assert torch.equal(a, b), i
        for attr in ("_prep", "_prep_up", "_prep_pool", "_prep_res", "_prep_cpres"):
            pa, pb = getattr(a, attr, None), getattr(b, attr, None)
            assert (pa is None) == (pb is None), (i, attr)
            if pa is not None:
                for x, y in zip(pa, pb):
                    assert (x is None) == (y is None) and (x is None or torch.equal(x.view(torch.int16), y.view(torch.int16))), (i, attr)
    # the table on its own, from the normalised weight: embedding_fwd + linear_fwd bit for bit
    assert torch.equal(K.label_dense_table(table, Wb_ref[3], bias).view(torch.int16), T_ref.view(torch.int16))
    # from the master weight and sigma (what the fused launch does): the sum is divided ONCE instead of every weight -- the same
    # value up to one rounding of the bf16 result
    sg = fused.scal[8 * 3:8 * 3 + 1]
    t64 = (table.to(torch.bfloat16).double() @ (Ws[3].double() / sg.double()) + bias.double()).cpu().numpy()
    for T in (Wb[3]._label_T, K.label_dense_table(table, Ws[3], bias, sigma=sg)):
        assert relerr(T, t64) < 2.0 ** -8
        assert float((T.float() - T_ref.float()).abs().max()) <= 2.0 ** -8 * float(T_ref.float().abs().max())


@pytest.mark.parametrize("kind,loss_type,critic", [(5, "Goodfellow", True), (6, "Goodfellow", False), (7, "HINGE", True), (2, "WGAN", True),
                                                   (3, "HINGE", False), (3, "WGAN", False)])
def test_soft_plus_variants_of_the_sngan_losses(K, kind, loss_type, critic):
    """SOFT_PLUS = True (SNGAN/gan_cifar_resnet.py:63) wraps the three loss types in softplus (:364-386 critic, :483-497 generator):
    gank_gan_pointwise_loss kinds 5 / 6 / 7, and kinds 2 / 3 for the branches that coincide with the sigmoid cross-entropy and its
    non-saturating generator.  Loss and d loss / d logits against torch-float64 autograd of the script's expressions
    (oracle.ref_torch.sngan_losses) on the same bf16 logits, incl. logits far outside (-1, 1) on both sides."""
    from oracle import ref_torch as T
    rng = np.random.default_rng(kind * 7 + len(loss_type))
    n, b = 96, 40
    lg = np.concatenate([rng.normal(size=n - 8) * 2.0, [-9.0, -3.5, -1.25, -0.75, 0.75, 1.25, 3.5, 9.0]])
    rng.shuffle(lg)
    lt = torch.tensor(lg, dtype=torch.float32).to(torch.bfloat16).cuda()
    x = lt.double().cpu().requires_grad_(True)
    d, g = T.sngan_losses(x if critic else None, b, None if critic else x, loss_type, True)
    ref = d if critic else g
    (rg,) = torch.autograd.grad(ref, [x])
    loss, dl, dl32 = K.gan_pointwise_loss(lt, b if critic else 0, kind)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    assert relerr(dl32, rg.numpy()) < 1e-5 and relerr(dl, rg.numpy()) < BF_TOL


@pytest.mark.parametrize("n", [6, 128])
def test_conv3x3_with_its_spatially_constant_input_channels_factored_out(K, n):
    """csrc/label_conv.hip (round 5): D.Block.2.Conv1 (SNGAN/gan_cifar_resnet.py:186-190) reads concat(features, tile(label vector))
    (:282-284), so half of its input channels hold one vector per sample.  gank_label_conv3x3_table + gank_img16_conv3x3_label_bias
    (forward), the image-resident input gradient on the feature half, gank_conv2d_wgrad on the feature half + gank_label_conv3x3_bwd
    (filter gradient of both halves, gradient of the tiled vector) against float64 of the UNFACTORED layer on the concatenated
    tensor: y, dW (all 256 input channels), the feature gradient and the per-sample gradient of the tiled vector."""
    rng = np.random.default_rng(91 + n)
    c1 = c2 = 128
    cout, v = 256, 10
    a, at = bf(rng.normal(size=(n, 16, 16, c1)))
    tab, tt = bf(rng.normal(size=(v, c2)) * 0.7)
    labels = rng.integers(0, v, n)
    lt = torch.tensor(labels, dtype=torch.int32).cuda()
    w, _ = bf(rng.normal(size=(3, 3, c1 + c2, cout)) / np.sqrt(9 * (c1 + c2)))       # bf16-representable: the operands round nothing away
    bias, bt = f32(rng.normal(size=cout) * 0.1)
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    xcat = np.concatenate([a, np.broadcast_to(tab[labels][:, None, None, :], (n, 16, 16, c2))], axis=3)
    y_ref = R.conv2d_same(R.relu(xcat), w) + bias
    (rf, rd), = K.prep_weights_batched([wt], want_d=True, kinds=[6])
    table, lists = K.label_conv3x3_table(wt, c1, tt, bt, lt)
    assert torch.equal(K.label_conv3x3_table(wt, c1, tt, bt), table)
    for lab in range(v):             # row v of the lists: {count, the samples of the label in ascending order}
        idx = np.nonzero(labels == lab)[0]
        row = lists[lab].cpu().numpy()
        assert row[0] == len(idx) and np.array_equal(row[1:1 + len(idx)], idx)
    y = K.img16_conv3x3_label_bias(at, rf, table, lt, cout, K.IN_RELU)
    torch.cuda.synchronize()
    assert relerr(y, y_ref) < BF_TOL, relerr(y, y_ref)
    # backward
    dy, dyt = bf(rng.normal(size=(n, 16, 16, cout)))
    dx_ref, dw_ref, _ = R.conv2d_same_grads(R.relu(xcat), w, dy)
    dx_ref = dx_ref * (xcat > 0)
    da_ref, de_ref = dx_ref[..., :c1], dx_ref[..., c1:].sum((1, 2))
    dw0 = rng.normal(size=w.shape).astype(np.float32)              # accumulation into a non-zero gradient buffer
    dw = torch.tensor(dw0).cuda()
    dwf = torch.zeros((3, 3, c1, cout), device="cuda")
    jobs = []
    K.conv2d_wgrad(at, dyt, dwf, (16, 16), 3, K.IN_RELU, slab_jobs=jobs)
    if jobs:
        K.sum_slabs(jobs)
    parts = K.label_conv3x3_bwd(dyt, lists, tt, wt, c1, dw, dwf)
    da = K.img16_conv3x3(dyt, rd, None, c1, 0, relu_ref=at)
    torch.cuda.synchronize()
    assert float(dwf.abs().max()) == 0.0                           # (left clean for the next pass)
    assert relerr(dw - torch.tensor(dw0).cuda(), dw_ref) < F32_FROM_BF_TOL, relerr(dw - torch.tensor(dw0).cuda(), dw_ref)
    # (the vector gradient leaves summed per label: [9 taps][V][C2])
    de = parts.sum(0).double().cpu().numpy()
    for lab in range(v):
        idx = np.nonzero(labels == lab)[0]
        if len(idx):
            assert relerr(de[lab], de_ref[idx].sum(0)) < F32_FROM_BF_TOL, (lab, relerr(de[lab], de_ref[idx].sum(0)))
        else:
            assert float(np.abs(de[lab]).max()) == 0.0
    assert relerr(da, da_ref) < BF_TOL
    # the feature half's filter gradient straight into its rows of the full gradient (a strided slab job: no staging buffer)
    if K.conv2d_wgrad_rows_ok(n, (16, 16), c1, cout):
        dw2 = torch.tensor(dw0).cuda()
        jobs = []
        K.conv2d_wgrad_rows(at, dyt, dw2, (16, 16), 3, K.IN_RELU, jobs)
        assert len(jobs) == 1 and torch.equal(dw2.cpu(), torch.tensor(dw0))
        K.sum_slabs(jobs)
        K.label_conv3x3_bwd(dyt, lists, tt, wt, c1, dw2)
        torch.cuda.synchronize()
        assert relerr(dw2 - torch.tensor(dw0).cuda(), dw_ref) < F32_FROM_BF_TOL
        # ... and with the per-label tap sums computed by extra workgroups of that launch: bit-identical results
        dw3 = torch.tensor(dw0).cuda()
        sums = K.conv2d_wgrad_rows(at, dyt, dw3, (16, 16), 3, K.IN_RELU, jobs, tap_sums=(lists, v))
        K.sum_slabs(jobs)
        parts3 = K.label_conv3x3_bwd(dyt, lists, tt, wt, c1, dw3, sums=sums)
        torch.cuda.synchronize()
        assert torch.equal(dw3, dw2) and torch.equal(parts3, parts)
        # ... and with the label gradients as extra workgroups of the input-gradient launch
        dw4 = torch.tensor(dw0).cuda()
        sums = K.conv2d_wgrad_rows(at, dyt, dw4, (16, 16), 3, K.IN_RELU, jobs, tap_sums=(lists, v))
        K.sum_slabs(jobs)
        da4, parts4 = K.img16_conv3x3_label_bwd(dyt, rd, at, c1, sums, tt, wt, c1, dw4)
        torch.cuda.synchronize()
        assert torch.equal(dw4, dw2) and torch.equal(parts4, parts) and torch.equal(da4.view(torch.int16), da.view(torch.int16))
        # ... and with the pooled shortcut branch's share of the tiled vector's gradient as a tenth part, summed per label
        gq, gqt = bf(np.random.default_rng(7).normal(size=(n, 8, 8, c1 + c2)))
        dw5 = torch.tensor(dw0).cuda()
        sums = K.conv2d_wgrad_rows(at, dyt, dw5, (16, 16), 3, K.IN_RELU, jobs, tap_sums=(lists, v))
        K.sum_slabs(jobs)
        parts10 = K.label_conv3x3_bwd_pooled(sums, lists, tt, wt, c1, dw5, gqt, c1, n)
        torch.cuda.synchronize()
        assert torch.equal(dw5, dw2) and torch.equal(parts10[:9], parts)
        want9 = np.zeros((v, c2))
        for i in range(n):
            want9[labels[i]] += gq[i, :, :, c1:].reshape(64, c2).sum(0)
        assert relerr(parts10[9], want9) < 1e-6
        # ... and that launch as extra workgroups of the slab-summing launch (gank_sum_slabs_label_bwd): the same bits
        dw6 = torch.tensor(dw0).cuda()
        sums = K.conv2d_wgrad_rows(at, dyt, dw6, (16, 16), 3, K.IN_RELU, jobs, tap_sums=(lists, v))
        parts11 = K.sum_slabs(jobs, (sums, lists, tt, wt, c1, dw6, gqt, c1, n))
        torch.cuda.synchronize()
        assert len(jobs) == 0 and torch.equal(dw6, dw2) and torch.equal(parts11, parts10)
    else:
        assert n < 64
    # the pooled / unpooled ends of the pair: the pooled concat alone, and the gradient join with the factored consumer's partial sums
    _, yp = K.concat_label_pool_fwd(at, tt, lt, want_full=False)
    y_full, yp_full = K.concat_label_pool_fwd(at, tt, lt)
    table2, lists2, yp2 = K.label_conv3x3_table_pooled(wt, c1, tt, bt, lt, at)          # the table and the pooled concatenation in one launch
    torch.cuda.synchronize()
    assert torch.equal(table2, table) and torch.equal(yp2.view(torch.int16), yp_full.view(torch.int16))
    for lab in range(v):
        cnt = int(lists[lab, 0])
        assert torch.equal(lists2[lab, :1 + cnt], lists[lab, :1 + cnt])
    # ... and with the block's 1x1 shortcut conv on the pooled concatenation computed by that launch (one workgroup per sample)
    wsn, wst = f32(np.random.default_rng(11).normal(size=(1, 1, c1 + c2, 128)) * 0.1)
    bsn, bst = f32(np.random.default_rng(12).normal(size=128) * 0.1)
    wsf, _ = K.prep_weights(wst, True, False)
    assert K.label_conv3x3_table_pooled_shortcut_ok(at, c2, wsf, 128)
    table3, lists3, yp3, sc3 = K.label_conv3x3_table_pooled(wt, c1, tt, bt, lt, at, (wsf, bst, 128))
    sc_launch = K.conv2d_fprop(yp_full, wsf, bst, (8, 8), 128, 1)
    torch.cuda.synchronize()
    assert torch.equal(table3, table) and torch.equal(yp3.view(torch.int16), yp_full.view(torch.int16))
    sc_ref = yp_full.double().cpu().numpy().reshape(-1, c1 + c2) @ torch.tensor(wsn).to(torch.bfloat16).double().numpy().reshape(c1 + c2, 128) + bsn
    assert relerr(sc3, sc_ref.reshape(n, 8, 8, 128)) < BF_TOL
    assert float((sc3.float() - sc_launch.float()).abs().max()) <= 2.0 ** -7 * float(sc_launch.float().abs().max())     # (another summation order)
    gp, gpt = bf(rng.normal(size=(n, 8, 8, c1 + c2)))
    da2, de2 = K.concat_label_unpool_bwd_factored(da, gpt, parts, lt, lists)
    gm_full = torch.cat([da, torch.zeros((n, 16, 16, c2), dtype=da.dtype, device="cuda")], 3).contiguous()
    da3, de3 = K.concat_label_unpool_bwd(gm_full, gpt, c1)
    torch.cuda.synchronize()
    assert torch.equal(yp.view(torch.int16), yp_full.view(torch.int16)) and torch.equal(da2.view(torch.int16), da3.view(torch.int16))
    # the feature half's join inside the input-gradient launch: relu_mask(conv(dy)) + 0.25 * unpool(gp[..., :C1]) with ONE rounding
    da5 = K.img16_conv3x3_dgrad_unpool(dyt, rd, at, gpt, c1, 0.25)
    _, de5 = K.concat_label_unpool_bwd_factored(c1, gpt, parts, lt, lists)
    torch.cuda.synchronize()
    assert relerr(da5, da_ref + 0.25 * np.repeat(np.repeat(gp[..., :c1], 2, axis=1), 2, axis=2)) < BF_TOL
    assert float((da5.float() - da3.float()).abs().max()) <= 2.0 ** -7 * float(da3.float().abs().max())        # (the two-pass form rounds twice)
    assert relerr(de5, de2.double().cpu().numpy()) < 1e-6           # (the sum of the pooled gradient itself: another order of the same additions)
    want = de3.double().cpu().numpy().copy()            # the per-label sums join the row of the label's first sample
    for lab in range(v):
        idx = np.nonzero(labels == lab)[0]
        if len(idx):
            want[idx[0]] += de[lab]
    assert relerr(de2, want) < 1e-6


def test_concat_rows_is_two_copies_and_its_backward_two_views(K):
    """functional.concat_rows (tf.concat(axis=0) of the real and fake logits in front of a critic loss): the library's copy
    kernel twice; the gradient comes back as the two row ranges"""
    from gan_lib_tensorflow_amd import functional as Fn
    a = torch.randn(5, 3, device="cuda").to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(7, 3, device="cuda").to(torch.bfloat16).requires_grad_(True)
    y = Fn.concat_rows(a, b)
    assert torch.equal(y.detach(), torch.cat([a.detach(), b.detach()], 0))
    g = torch.randn(12, 3, device="cuda").to(torch.bfloat16)
    y.backward(g)
    assert torch.equal(a.grad, g[:5]) and torch.equal(b.grad, g[5:])


def test_weighted_sum_of_loss_terms_and_its_seeds(K):
    """functional.weighted_sum (gen_loss = gan_weight * GAN + l1_weight * L1; d_loss = gan + gp + ac): one launch forward; backward
    hands a unit seed through weight 1 untouched (identity), turns it into the persistent constant seed of any other weight, and
    scales an ordinary upstream gradient on the device -- the gradients of the terms are w_i * g in all three cases."""
    from gan_lib_tensorflow_amd import functional as Fn
    a = torch.tensor([1.5], device="cuda", requires_grad=True)
    b = torch.tensor([-0.25], device="cuda", requires_grad=True)
    c = torch.tensor([4.0], device="cuda", requires_grad=True)
    y = Fn.weighted_sum([a, b, c], [1.0, 100.0, 0.5])
    assert abs(float(y) - (1.5 - 25.0 + 2.0)) < 1e-6
    seed = Fn.unit_seed(y)
    ga, gb, gc = torch.autograd.grad([y], [a, b, c], [seed], retain_graph=True)
    assert ga.data_ptr() == seed.data_ptr() and float(gb) == 100.0 and float(gc) == 0.5
    assert Fn._seed_scale.get(gb.data_ptr()) == 100.0
    ga, gb, gc = torch.autograd.grad([y], [a, b, c], [torch.tensor([3.0], device="cuda")])
    assert (float(ga), float(gb), float(gc)) == (3.0, 300.0, 1.5)
    acc = torch.full((7, 5), 2.0, device="cuda")
    K.weighted_sum_f32([acc, torch.ones((7, 5), device="cuda")], [1.0, 1.0], out=acc)      # in-place accumulation
    assert bool((acc == 3.0).all())


def test_concat_label_fwd_bwd(K):
    """The critic's label branch through the per-label table: forward = concat_tile(x, linear(embedding(labels))) bit for
    bit (table built from the normalised weight); backward against float64 (the fp32 per-label sums are more accurate than the per-sample bf16 chain they replace)."""
    rng = np.random.default_rng(52)
    n, hw, c1, c2, v, d = 24, 16, 128, 128, 10, 300
    a, at = bf(rng.normal(size=(n, 4, 4, c1)))
    labels = rng.integers(0, v, n)
    lt = torch.tensor(labels, dtype=torch.int32).cuda()
    tab, tabt = f32(rng.uniform(-0.08, 0.08, size=(v, d)))
    w, wt = f32(rng.normal(size=(d, c2)) * 0.05)
    b, bt = f32(rng.normal(size=c2) * 0.1)
    T = K.label_dense_table(tabt, wt, bt)
    y = K.concat_label_fwd(at, T, lt)
    y_ref = K.concat_tile_fwd(at, K.linear_fwd(K.embedding_fwd(tabt, lt), wt, bt))
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    dy, dyt = bf(rng.normal(size=(n, 4, 4, c1 + c2)))
    da, de32 = K.concat_label_bwd(dyt, c1)
    assert torch.equal(da.view(torch.int16), dyt[..., :c1].contiguous().view(torch.int16))
    de = dy[..., c1:].reshape(n, hw, c2).sum(1)
    assert relerr(de32, de) < 1e-6
    dw, db, dtab = torch.zeros_like(wt), torch.zeros_like(bt), torch.zeros_like(tabt)
    K.label_dense_bwd(de32, lt, tabt, wt, dw, db, dtab)
    K.label_dense_bwd(de32, lt, tabt, wt, dw, db, dtab)          # accumulates
    torch.cuda.synchronize()
    dT = np.zeros((v, c2))
    for i in range(n):
        dT[labels[i]] += de[i]
    E = torch.tensor(tab, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    assert relerr(dw, 2 * E.T @ dT) < 1e-5
    assert relerr(db, 2 * dT.sum(0)) < 1e-5
    assert relerr(dtab, 2 * dT @ w.T) < 1e-5
    # deterministic: a second pair of launches from zero gives the same bits
    dw2, db2, dtab2 = torch.zeros_like(wt), torch.zeros_like(bt), torch.zeros_like(tabt)
    K.label_dense_bwd(de32, lt, tabt, wt, dw2, db2, dtab2)
    K.label_dense_bwd(de32, lt, tabt, wt, dw2, db2, dtab2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dtab, dtab2)
    # the same from rows that are summed per label already (gank_label_dense_bwd_parts: dT[l] = sum_p parts[p][l])
    pr, prt = f32(rng.normal(size=(10, v, c2)))
    dw3, db3, dtab3 = torch.zeros_like(wt), torch.zeros_like(bt), torch.zeros_like(tabt)
    K.label_dense_bwd_parts(prt, tabt, wt, dw3, db3, dtab3)
    K.label_dense_bwd_parts(prt, tabt, wt, dw3, db3, dtab3)
    torch.cuda.synchronize()
    dT3 = pr.astype(np.float64).sum(0)
    assert relerr(dw3, 2 * E.T @ dT3) < 1e-5 and relerr(db3, 2 * dT3.sum(0)) < 1e-5 and relerr(dtab3, 2 * dT3 @ w.T) < 1e-5
    # a label outside the table: a zero row forward, no contribution backward
    lt2 = lt.clone(); lt2[0] = 99
    y2 = K.concat_label_fwd(at, T, lt2)
    assert float(y2[0, :, :, c1:].abs().max()) == 0.0 and torch.equal(y2[1:].view(torch.int16), y[1:].view(torch.int16))


@pytest.mark.parametrize("n,hw,cin,cout", [(128, 8, 256, 128), (8, 8, 64, 64), (64, 4, 128, 128)])
def test_conv1x1_both_gradients_in_one_launch(K, n, hw, cin, cout):
    """gank_conv1x1_wgrad_dgrad (the backward of D.Block.2.Shortcut in a critic update, common/ops/conv2d.py:180-187 with a 1x1 filter):
    the filter / bias gradient of gank_conv2d_wgrad and the input gradient of gank_conv2d_dgrad, the latter from extra workgroups of the
    former's launch -- against float64 and against the two launches."""
    rng = np.random.default_rng(61)
    x, xt = bf(rng.normal(size=(n, hw, hw, cin)))
    dy, dyt = bf(rng.normal(size=(n, hw, hw, cout)))
    w, wt = f32(rng.normal(size=(1, 1, cin, cout)) * 0.1)
    _, wd = K.prep_weights(wt, False, True)
    assert K.conv1x1_wgrad_dgrad_ok(n, (hw, hw), cin, cout, wd)
    dw, db = torch.zeros_like(wt), torch.zeros(cout, dtype=torch.float32, device="cuda")
    dx = K.conv1x1_wgrad_dgrad(xt, dyt, dw, wd, dbias=db)
    dw2, db2 = torch.zeros_like(wt), torch.zeros(cout, dtype=torch.float32, device="cuda")
    K.conv2d_wgrad(xt, dyt, dw2, (hw, hw), 1, 0, 1.0, dbias=db2)
    dx2 = K.conv2d_dgrad(dyt, wd, (hw, hw), cin, 1)
    torch.cuda.synchronize()
    X, DY = x.reshape(-1, cin).astype(np.float64), dy.reshape(-1, cout).astype(np.float64)
    wb = torch.tensor(w).to(torch.bfloat16).double().numpy().reshape(cin, cout)
    assert relerr(dw, (X.T @ DY).reshape(1, 1, cin, cout)) < F32_FROM_BF_TOL and relerr(db, DY.sum(0)) < F32_FROM_BF_TOL
    assert relerr(dx, (DY @ wb.T).reshape(n, hw, hw, cin)) < BF_TOL
    assert relerr(dw, dw2.double().cpu().numpy()) < 1e-5 and relerr(db, db2.double().cpu().numpy()) < 1e-5      # (fp32 atomics: the order is free)
    assert float((dx.float() - dx2.float()).abs().max()) <= 2.0 ** -7 * float(dx2.float().abs().max())          # (another summation order)


def test_meanpool_conv1x1_gather_is_pool_then_conv(K):
    """D.Block.1.Shortcut with the 2x2 mean inside the conv's gather: the bytes of pool2x2 + conv2d_fprop, and the pooled image
    as a side output."""
    rng = np.random.default_rng(53)
    for n in (3, 128):
        x, xt = bf(rng.normal(size=(n, 32, 32, 3)))
        w, wt = f32(rng.normal(size=(1, 1, 3, 128)) * 0.3)
        b, bt = f32(rng.normal(size=128) * 0.1)
        wf, _ = K.prep_weights(wt, True, False)
        pooled_ref = K.pool2x2(xt, 0.25)
        y_ref = K.conv2d_fprop(pooled_ref, wf, bt, (16, 16), 128, 1)
        y, pooled = K.meanpool_conv1x1_fprop(xt, wf, bt, 128)
        torch.cuda.synchronize()
        assert torch.equal(pooled.view(torch.int16), pooled_ref.view(torch.int16))
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
        assert relerr(y, R.conv2d_same(R.meanpool2x2(x), w) + b) < BF_TOL
        # ... and with the block's 3x3 conv on the same image in the same launch (gank_image_conv_pair_fprop): the bytes of the two entries
        w1, w1t = f32(rng.normal(size=(3, 3, 3, 128)) * 0.2)
        b1, b1t = f32(rng.normal(size=128) * 0.1)
        wf1, _ = K.prep_weights(w1t, True, False)
        y1_ref = K.conv2d_fprop(xt, wf1, b1t, (32, 32), 128, 3)
        y1, ys, pooled2 = K.image_conv_pair_fprop(xt, wf1, b1t, 128, wf, bt, 128)
        torch.cuda.synchronize()
        assert torch.equal(y1.view(torch.int16), y1_ref.view(torch.int16)) and torch.equal(ys.view(torch.int16), y.view(torch.int16))
        assert torch.equal(pooled2.view(torch.int16), pooled.view(torch.int16))
        assert relerr(y1, R.conv2d_same(x, w1) + b1) < BF_TOL


def test_narrow_input_filter_gradient_streaming_kernel(K):
    """Filter gradients of 3-channel-input layers on the streaming kernel (the whole dy slice by LDS-DMA, the bias gradient as an
    all-ones operand row): single layers through gank_conv2d_wgrad, the critic's pair (3x3 at 32x32 + 1x1 at 16x16) in one
    launch, ragged pixel counts (M not a multiple of 512), accumulation into non-zero buffers."""
    rng = np.random.default_rng(54)
    cases = [(128, 32, 3, 128), (128, 16, 1, 128), (5, 32, 3, 256), (3, 16, 1, 128), (7, 8, 3, 128), (2, 4, 3, 128)]
    data = []
    for n, hw, k, cout in cases:
        x, xt = bf(rng.normal(size=(n, hw, hw, 3)))
        dy, dyt = bf(rng.normal(size=(n, hw, hw, cout)))
        w0 = rng.normal(size=(k, k, 3, cout)).astype(np.float32)
        b0 = rng.normal(size=cout).astype(np.float32)
        if k == 3:
            _, rdw, rdb = R.conv2d_same_grads(x, np.zeros((3, 3, 3, cout)), dy)
        else:
            rdw = np.einsum('nhwc,nhwo->co', x, dy).reshape(1, 1, 3, cout)
            rdb = dy.sum((0, 1, 2))
        dw, db = torch.tensor(w0).cuda(), torch.tensor(b0).cuda()
        K.conv2d_wgrad(xt, dyt, dw, (hw, hw), k, 0, 1.0, dbias=db)
        torch.cuda.synchronize()
        assert relerr(dw - torch.tensor(w0).cuda(), rdw) < F32_FROM_BF_TOL, (n, hw, k, cout)
        assert relerr(db - torch.tensor(b0).cuda(), rdb) < F32_FROM_BF_TOL, (n, hw, k, cout)
        data.append((xt, dyt, hw, k, cout, rdw, rdb))
    for i, j in ((0, 1), (1, 0), (2, 3), (4, 5)):
        outs = []
        items = []
        for xt, dyt, hw, k, cout, rdw, rdb in (data[i], data[j]):
            dw, db = torch.zeros((k, k, 3, cout), device="cuda"), torch.zeros(cout, device="cuda")
            outs.append((dw, db, rdw, rdb))
            items.append((xt, dyt, dw, db, (hw, hw), k))
        K.conv2d_wgrad_narrow_pair(items[0], items[1])
        torch.cuda.synchronize()
        for dw, db, rdw, rdb in outs:
            assert relerr(dw, rdw) < F32_FROM_BF_TOL and relerr(db, rdb) < F32_FROM_BF_TOL, (i, j)


def test_concat_label_with_the_next_blocks_fan_out(K):
    """concat + fork_pool of the down-sampling block behind it as one launch each way: the bytes of the separate launches
    (concat_label_fwd, pool2x2; unpool2x2_add, concat_label_bwd), the per-sample sums up to fp32 summation order."""
    rng = np.random.default_rng(55)
    n, c1, c2, v = 6, 128, 128, 10
    a, at = bf(rng.normal(size=(n, 16, 16, c1)))
    T, Tt = bf(rng.normal(size=(v, c2)))
    lt = torch.tensor(rng.integers(0, v, n), dtype=torch.int32).cuda()
    y, yp = K.concat_label_pool_fwd(at, Tt, lt)
    y_ref = K.concat_label_fwd(at, Tt, lt)
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
    assert torch.equal(yp.view(torch.int16), K.pool2x2(y_ref, 0.25).view(torch.int16))
    gm, gmt = bf(rng.normal(size=(n, 16, 16, c1 + c2)))
    gp, gpt = bf(rng.normal(size=(n, 8, 8, c1 + c2)))
    da, de = K.concat_label_unpool_bwd(gmt, gpt, c1)
    da_ref, de_ref = K.concat_label_bwd(K.unpool2x2_add(gpt, gmt, 0.25), c1)
    torch.cuda.synchronize()
    assert torch.equal(da.view(torch.int16), da_ref.view(torch.int16))
    assert relerr(de, de_ref.double().cpu().numpy()) < 1e-6
    da2, de2 = K.concat_label_unpool_bwd(None, gpt, c1)
    da2_ref, de2_ref = K.concat_label_bwd(K.unpool2x2_add(gpt, None, 0.25), c1)
    assert torch.equal(da2.view(torch.int16), da2_ref.view(torch.int16)) and relerr(de2, de2_ref.double().cpu().numpy()) < 1e-6


@pytest.mark.parametrize("n,hp,cin,shortcut,bias,slabs", [(3, 16, 128, True, True, False), (2, 8, 128, False, True, True), (5, 16, 256, True, False, True),
                                                         (128, 16, 128, True, True, False), (128, 16, 128, True, True, True), (3, 16, 128, True, False, True)])
def test_convpool_input_gradient_with_the_image_convs_filter_gradient_inside(K, n, hp, cin, shortcut, bias, slabs):
    """gank_cpool_res_dgrad_image_wgrad (round 5): the ConvMeanPool input gradient of OptimizedResBlockDisc1 in a critic update
    (gan_cifar_resnet.py:212-234) never stores its result; the launch accumulates the filter / bias gradient of the 3x3 conv on the
    3-channel image in front (dw1, db1) and of the 1x1 shortcut conv on the pooled image (dws, dbs).  Oracle: float64 gradients of
    the un-reduced composition, fed the tensor the UNFUSED path stores (the oracle's masked input gradient rounded to bf16 --
    the fused kernel rounds its tiles the same way), tolerance of an fp32 sum of bf16 products; and against the two-launch path
    (gank_cpool_res_dgrad + gank_conv2d_wgrad_narrow_pair) on the same inputs; accumulation into non-zero buffers."""
    wp, cout = 16, 128
    rng = np.random.default_rng(n * 7 + hp + cin)
    img, imgt = bf(rng.normal(size=(n, 2 * hp, 2 * wp, 3)))
    h1, h1t = bf(rng.normal(size=(n, 2 * hp, 2 * wp, cin)))                  # Conv1's output = the relu reference
    w2, _ = bf(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    dy, dyt = bf(rng.normal(size=(n, hp, wp, cout)))
    pooled = R.meanpool2x2(img)
    pooled, pooledt = bf(pooled)
    w2t = torch.tensor(w2, dtype=torch.float32).cuda()
    (rf, rd), = K.prep_weights_batched([w2t], want_d=True, kinds=[5])
    assert K.cpool_res_dgrad_image_wgrad_ok(dyt, cin)
    # reference: dh = mask * dgrad(ConvMeanPool), rounded as the stored tensor; then the two filter gradients
    dh_ref, _, _ = R.conv2d_same_grads(R.relu(h1), w2, R.meanpool2x2_grad(dy))
    dh_ref = dh_ref * (h1 > 0)
    dh_b = torch.tensor(dh_ref, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()
    _, rdw1, rdb1 = R.conv2d_same_grads(img, np.zeros((3, 3, 3, cin)), dh_b)
    rdws = np.einsum('nhwc,nhwo->co', pooled, dy).reshape(1, 1, 3, cout)
    rdbs = dy.sum((0, 1, 2))
    w0, b0 = rng.normal(size=(3, 3, 3, cin)).astype(np.float32), rng.normal(size=cin).astype(np.float32)       # non-zero targets
    ws0, bs0 = rng.normal(size=(1, 1, 3, cout)).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    dw1, db1 = torch.tensor(w0).cuda(), torch.tensor(b0).cuda()
    dws, dbs = torch.tensor(ws0).cuda(), torch.tensor(bs0).cuda()
    jobs = [] if slabs else None       # slabs: each workgroup's tile to a slab of its own (Cin == 128; else the atomics), summed by ONE later launch
    K.cpool_res_dgrad_image_wgrad(dyt, rd, h1t, imgt, dw1, db1 if bias else None, pooledt if shortcut else None,
                                  dws if shortcut else None, dbs if (shortcut and bias) else None, slab_jobs=jobs)
    if slabs and cin == 128:
        assert len(jobs) == 1 + int(bias) + (1 + int(bias)) * int(shortcut) and torch.equal(dw1.cpu(), torch.tensor(w0))
        K.sum_slabs(jobs)
    elif slabs:
        assert jobs == []
    torch.cuda.synchronize()
    # a few tiles of dh differ from the oracle's rounding by one bf16 ulp where the fp32 sum sits on a rounding boundary: the bound
    # is the bf16-output one on the sum's scale, not the fp32-from-identical-bf16 one
    assert relerr(dw1 - torch.tensor(w0).cuda(), rdw1) < 4e-3, relerr(dw1 - torch.tensor(w0).cuda(), rdw1)
    if bias:
        assert relerr(db1 - torch.tensor(b0).cuda(), rdb1) < 4e-3
    else:
        assert torch.equal(db1.cpu(), torch.tensor(b0))
    if shortcut:
        assert relerr(dws - torch.tensor(ws0).cuda(), rdws) < F32_FROM_BF_TOL
        if bias:
            assert relerr(dbs - torch.tensor(bs0).cuda(), rdbs) < F32_FROM_BF_TOL
    else:
        assert torch.equal(dws.cpu(), torch.tensor(ws0)) and torch.equal(dbs.cpu(), torch.tensor(bs0))
    # the two-launch path on the same inputs
    dx = K.cpool_res_dgrad(dyt, rd, cin, h1t)
    u1, ub1 = torch.zeros((3, 3, 3, cin), device="cuda"), torch.zeros(cin, device="cuda")
    us, ubs = torch.zeros((1, 1, 3, cout), device="cuda"), torch.zeros(cout, device="cuda")
    K.conv2d_wgrad(imgt, dx, u1, (2 * hp, 2 * wp), 3, 0, 1.0, dbias=ub1)
    K.conv2d_wgrad(pooledt, dyt, us, (hp, wp), 1, 0, 1.0, dbias=ubs)
    torch.cuda.synchronize()
    assert relerr(dw1 - torch.tensor(w0).cuda(), u1.double().cpu().numpy()) < 1e-3           # same bf16 operands, other summation order
    if bias:
        assert relerr(db1 - torch.tensor(b0).cuda(), ub1.double().cpu().numpy()) < 1e-3
    if shortcut:
        assert relerr(dws - torch.tensor(ws0).cuda(), us.double().cpu().numpy()) < 1e-3


@pytest.mark.parametrize("health,dw_zero", [(False, False), (True, False), (False, True)])
def test_spectral_norm_backward_adam_and_next_power_iteration_in_one_launch(K, health, dw_zero):
    """gank_sn_adam_fwd_a (round 5): the end of a critic update -- spectral-norm backward apply, TF-Adam over the flat buffer, and
    the NEXT forward pass's power iteration on the updated weights -- as one launch, against the launches it replaces
    (gank_sn_power_iter_bwd, gank_adam_tf_health with the gradient clear, gank_sn_power_iter_fwd_a): BIT-IDENTICAL parameters,
    Adam slots, cleared gradients, workspaces (a, b, scal) and staged u'; then the consumer side: a forward pass that runs its
    second launch only (gank_sn_power_iter_fwd_b_prep) and adopts u' equals the two-launch forward pass bit for bit.  Shapes:
    the critic's (27 x 128 image conv, 1152 x 128, 300 x 128 dense, 128 x 1 head behind an odd-sized bias: unaligned slices take
    the generic path), biases and an embedding table in the gaps."""
    torch.manual_seed(5)
    shapes = [(3, 3, 3, 128), (128,), (3, 3, 128, 128), (128,), (10, 300), (300, 128), (128,), (1, 1, 256, 128), (3, 3, 32, 256), (128, 1), (1,), (3, 3, 64, 64), (7,), (5, 5, 3, 32)]
    sn = [0, 2, 5, 7, 8, 9, 11, 13]
    sizes = [int(np.prod(sh)) for sh in shapes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    n = int(offs[-1])
    n_sn = sum(sizes[i] for i in sn)

    def build():
        g = torch.Generator().manual_seed(11)
        p = (torch.randn(n, generator=g) * 0.05).cuda()
        m, v = (torch.randn(n, generator=g) * 1e-3).cuda(), (torch.rand(n, generator=g) * 1e-4).cuda()
        grads_all = torch.zeros(n + n_sn + 8, device="cuda")
        grads = grads_all[:n]
        Ws = [p[offs[i]:offs[i] + sizes[i]].view(shapes[i]) for i in sn]
        dWs = [grads[offs[i]:offs[i] + sizes[i]].view(shapes[i]) for i in sn]
        cs = [shapes[i][-1] for i in sn]
        u_flat = torch.randn(sum(cs), generator=g).cuda()
        us, o = [], 0
        for c in cs:
            us.append(u_flat[o:o + c].view(1, c))
            o += c
        # gradients: plain ones for the gaps, dW_bar slices (scratch half) for the normalised weights, a little of both in dW
        for i in range(len(shapes)):
            if i not in sn:
                grads[offs[i]:offs[i] + sizes[i]] = torch.randn(sizes[i], generator=g).cuda() * 0.1
        Gs, o = [], n
        for i in sn:
            Gs.append(grads_all[o:o + sizes[i]].view(shapes[i]))
            Gs[-1].copy_((torch.randn(sizes[i], generator=g) * 0.1).view(shapes[i]))
            o += sizes[i]
        if not dw_zero:
            dWs[2].add_(0.01)
        if health:
            grads[offs[1] + 3] = float("inf")              # a bias gradient that overflowed
            Gs[1].view(-1)[17] = float("nan")             # ... and one element of a normalised weight's
            Gs[1].view(-1)[18] = 0.0
        hp = torch.tensor([2e-4, 0.0, 0.9, 1e-8, 0.5, 1.0, 0.0, 0.0], device="cuda")
        t = torch.tensor([3], dtype=torch.int64, device="cuda")
        it = torch.tensor([1234], dtype=torch.int64, device="cuda")
        hl = torch.zeros(2, dtype=torch.int64, device="cuda") if health else None
        st = K.SnState(Ws, us, u_flat)
        return dict(p=p, m=m, v=v, grads_all=grads_all, grads=grads, Ws=Ws, dWs=dWs, Gs=Gs, us=us, u_flat=u_flat, hp=hp, t=t, it=it, hl=hl, st=st)

    # ---- A: the launches it replaces
    A = build()
    bA = K.SnBatch(A["Ws"], A["us"], snapshot=True, inplace=True, state=A["st"])
    bA.forward()
    bA.backward(A["Gs"], A["dWs"])
    K.adam_tf(A["p"], A["grads_all"], A["m"], A["v"], A["hp"], A["t"], A["it"], zero_grads=True, health=A["hl"])
    A["st"].refresh()
    # ---- B: one launch
    B = build()
    bB = K.SnBatch(B["Ws"], B["us"], snapshot=True, inplace=True, state=B["st"])
    bB.forward()
    bB.backward_gw(B["Gs"], B["dWs"])
    bB.adam_fwd_a(B["p"], B["grads"], B["m"], B["v"], B["hp"], B["t"], B["it"], health=B["hl"], dw_zero=dw_zero)
    torch.cuda.synchronize()
    assert B["st"].valid
    for key in ("p", "m", "v", "grads_all", "u_flat", "t"):
        assert torch.equal(A[key], B[key]), key
    if health:
        # (the NaN in one dW_bar makes <dW_bar, W> and with it every gradient element of that weight non-finite: 3*3*128*128 + the bias's one)
        assert torch.equal(A["hl"], B["hl"]) and int(A["hl"][0]) == 3 * 3 * 128 * 128 + 1, (A["hl"], B["hl"])
        assert bool(torch.isfinite(B["p"]).all())               # ... and none of them touched p, m or v
    for name in ("a", "b", "scal", "u_next"):        # bit patterns: <dW_bar, W> of the poisoned weight is NaN in both
        assert torch.equal(getattr(A["st"], name).view(torch.int32), getattr(B["st"], name).view(torch.int32)), name
    assert float(B["grads_all"].abs().max()) == 0.0 and int(B["t"]) == 4
    assert float((B["p"] - build()["p"]).abs().max()) > 1e-5           # (the step did move the weights)
    # ---- consumer: second launch only + adoption of u' == the two-launch forward pass
    A["st"].valid = False
    fA = K.SnBatch(A["Ws"], A["us"], snapshot=True, inplace=True, state=A["st"])
    WA = fA.forward()
    fB = K.SnBatch(B["Ws"], B["us"], snapshot=True, inplace=True, state=B["st"])
    WB = fB.forward()
    torch.cuda.synchronize()
    assert not B["st"].valid
    for x, y in zip(WA, WB):
        assert torch.equal(x, y)
    for name in ("v", "ga", "u_snap"):
        assert torch.equal(getattr(A["st"], name), getattr(B["st"], name)), name
    assert torch.equal(A["u_flat"], B["u_flat"]) and torch.equal(B["u_flat"], B["st"].u_next)
    # ... and a reading pass (NO_OPS: u never written) on valid state
    B["st"].refresh()
    u_before = B["u_flat"].clone()
    rB = K.SnBatch(B["Ws"], B["us"], state=B["st"])
    WR = rB.forward()
    A["st"].valid = False
    rA = K.SnBatch(A["Ws"], A["us"])
    WRA = rA.forward()
    torch.cuda.synchronize()
    assert B["st"].valid and torch.equal(B["u_flat"], u_before)
    for x, y in zip(WRA, WR):
        assert torch.equal(x, y)


@pytest.mark.parametrize("n,h,cin,cout,k,relu", [(128, 16, 256, 256, 1, False), (128, 8, 256, 128, 1, False), (16, 32, 256, 3, 3, False), (8, 16, 64, 64, 1, True),
                                                 (4, 8, 128, 128, 3, False), (128, 16, 256, 256, 3, False)])
def test_conv_wgrad_split_slabs_instead_of_atomics(K, n, h, cin, cout, k, relu):
    """gank_conv2d_wgrad_slabs + gank_sum_slabs (round 5): the split-K kernels that add their partial tiles with fp32 atomics
    (1x1 shortcuts, the 256 -> 3 output layer's filter gradient, small per-tap layers) write per-split copies of the filter
    and ONE later launch sums them, scaled, into the non-zero target -- same result as the atomics form (oracle tolerance),
    bit-identical from run to run; a layer on the all-taps kernel (last case) leaves ITS slab reduction to the same launch."""
    rng = np.random.default_rng(n + h + cin + cout + k)
    x, xt = bf(rng.normal(size=(n, h, h, cin)))
    dy, dyt = bf(rng.normal(size=(n, h, h, cout)))
    xin = R.relu(x) if relu else x
    if k == 3:
        _, rdw, rdb = R.conv2d_same_grads(xin, np.zeros((3, 3, cin, cout)), dy)
    else:
        rdw, rdb = np.einsum('nhwc,nhwo->co', xin, dy).reshape(1, 1, cin, cout), dy.sum((0, 1, 2))
    w0 = rng.normal(size=(k, k, cin, cout)).astype(np.float32)
    flags = K.IN_RELU if relu else 0
    outs = []
    for rep in range(2):
        dw, db = torch.tensor(w0).cuda(), torch.zeros(cout, device="cuda")
        jobs = []
        K.conv2d_wgrad(xt, dyt, dw, (h, h), k, flags, 0.5, dbias=db, slab_jobs=jobs)
        expect_job = K.lib().gank_conv2d_wgrad_slab_elems(n, h, h, cin, cout, k, flags) > 0
        if not expect_job:
            assert jobs == []
        else:
            assert len(jobs) == 1 and torch.equal(dw.cpu(), torch.tensor(w0))          # nothing reached the target before the sum
            K.sum_slabs(jobs)
        torch.cuda.synchronize()
        assert relerr(dw - torch.tensor(w0).cuda(), 0.5 * rdw) < F32_FROM_BF_TOL
        assert relerr(db, 0.5 * rdb) < F32_FROM_BF_TOL
        outs.append((dw.clone(), expect_job))
    if (n, h, cin, cout, k) == (16, 32, 256, 3, 3):
        assert outs[0][1] and torch.equal(outs[0][0], outs[1][0])    # the output layer's geometry takes the slab form: deterministic


@pytest.mark.parametrize("n,hp,cin,cout", [(128, 8, 256, 128), (16, 16, 128, 128), (3, 8, 64, 64)])
def test_convpool_filter_gradient_fold_left_to_the_summing_launch(K, n, hp, cin, cout):
    """gank_convpool3x3_wgrad_job + gank_sum_slabs(fold = 1): the ConvMeanPool filter gradient's sum-and-fold (16 taps of the
    4x4 stride-2 form -> 9) as a job of the caller's ONE summing launch instead of a launch of its own: bit-identical to
    gank_convpool3x3_wgrad (the same additions in the same order), together with a plain job in the same launch."""
    rng = np.random.default_rng(n + hp + cin)
    _, xt = bf(rng.normal(size=(n, 2 * hp, 2 * hp, cin)))
    _, dyt = bf(rng.normal(size=(n, hp, hp, cout)))
    w0 = torch.tensor(rng.normal(size=(3, 3, cin, cout)).astype(np.float32)).cuda()
    a, da = w0.clone(), torch.zeros(cout, device="cuda")
    K.convpool3x3_wgrad(xt, dyt, a, K.IN_RELU, dbias=da)
    b, db = w0.clone(), torch.zeros(cout, device="cuda")
    jobs = []
    K.convpool3x3_wgrad(xt, dyt, b, K.IN_RELU, dbias=db, slab_jobs=jobs)
    extra_slabs, extra_out = torch.randn(3, 1000, device="cuda"), torch.zeros(1000, device="cuda")
    jobs.append(K.slab_job(extra_slabs, extra_out, 1000, 1000, 3, 2.0))
    assert len(jobs) == 2 and torch.equal(b, w0)
    K.sum_slabs(jobs)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert float((da - db).abs().max()) <= 1e-5 * float(da.abs().max())        # (the bias gradient is added with atomics in both forms: order noise)
    assert float((extra_out - 2.0 * extra_slabs.sum(0)).abs().max()) < 1e-5


@pytest.mark.parametrize("n,groups,res,stat_groups", [(6, 3, "half", 3), (4, 2, None, 0), (320, 10, "half", 10)])
def test_cbn_relu_fused_into_the_image_resident_16x16_conv(K, n, groups, res, stat_groups):
    """gank_cbn_relu_img16_conv3x3 (round 5): G.Block.2's  N2 -> relu -> Conv2 (+ half-resolution shortcut, + the statistics of the
    NEXT batch norm in the epilogue) as one launch for passes that keep nothing for a backward pass (gan_cifar_resnet.py:197-209;
    the 320-sample pass behind the critic updates) == cond-batch-norm launch + image-resident conv launch, BIT FOR BIT (the staged
    value is the forward CBN kernel's expression, rounded once), and against the float64 oracle."""
    rng = np.random.default_rng(n + groups)
    c = 256
    x, xt = bf(rng.normal(size=(n, 16, 16, c)) * 1.3 - 0.2)
    labels = torch.tensor(rng.integers(0, 10, n), dtype=torch.int32).cuda()
    gamma = torch.tensor(rng.normal(size=(10, c)) * 0.2 + 1, dtype=torch.float32).cuda()
    beta = torch.tensor(rng.normal(size=(10, c)) * 0.2, dtype=torch.float32).cuda()
    w, _ = bf(rng.normal(size=(3, 3, c, c)) / np.sqrt(9 * c))
    b, bt = f32(rng.normal(size=c))
    wt = torch.tensor(w, dtype=torch.float32).cuda()
    (rf, _), = K.prep_weights_batched([wt], want_d=True, kinds=[4])
    r8, r8t = bf(rng.normal(size=(n, 8, 8, c))) if res else (None, None)
    flags = K.RES_UPSAMPLE2X if res else 0
    yn, _ = K.cbn_fwd(xt, labels, gamma, beta, groups, True)
    stats = K.cbn_stats(xt, groups)
    if stat_groups:
        ref, cs_ref = K.img16_conv3x3(yn, rf, bt, c, flags, None, r8t, stat_groups)
        y, cs = K.cbn_relu_img16_conv3x3(xt, labels, gamma, beta, stats, rf, bt, c, flags, r8t, stat_groups)
    else:
        ref = K.img16_conv3x3(yn, rf, bt, c, flags, None, r8t)
        y = K.cbn_relu_img16_conv3x3(xt, labels, gamma, beta, stats, rf, bt, c, flags, r8t)
    torch.cuda.synchronize()
    if n * (c // 128) >= 256:
        assert torch.equal(y, ref)
    else:       # (small batches: the unfused conv takes the half-image form, whose two reduction halves add in another order)
        assert relerr(y, ref.double().cpu().numpy()) < 1e-2
    if stat_groups:         # the epilogue sums (atomics: order noise only)
        a, r_ = cs.sums.sum(1), cs_ref.sums.sum(1)
        assert float((a - r_).abs().max()) <= 1e-4 * float(r_.abs().max())
    if n <= 8:
        ry, _ = R.cond_batchnorm_forward(x, labels.cpu().numpy(), gamma.double().cpu().numpy(), beta.double().cpu().numpy(), groups)
        rr = R.conv2d_same(R.relu(ry), w, b)
        if res:
            rr = rr + np.repeat(np.repeat(r8, 2, axis=1), 2, axis=2)
        assert relerr(y, rr) < 2 * BF_TOL
