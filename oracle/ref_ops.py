"""NumPy float64 restatement of the reference op library, with hand-derived gradients.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).  Every function cites the
reference file:line (relative to /root/reference) it restates.  TensorFlow-1.5 op semantics that
the reference relies on (SAME padding, biased moments, depth_to_space channel order, TF Adam)
are written out explicitly because TensorFlow itself is not vendored in the reference.

Layouts follow the reference: activations NHWC, conv filters HWIO, linear weights [in, out].
"""
import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------------------
# tf.nn.conv2d, NHWC / HWIO, padding SAME        (call site: common/ops/conv2d.py:180-187)
# --------------------------------------------------------------------------------------
def same_pads(size, k, stride):
    """TF SAME rule: out = ceil(size/stride); the odd pad element goes AFTER (bottom/right)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, b=None, stride=1):
    """Cross-correlation (no filter flip), SAME padding.  x [N,H,W,Ci], w [kh,kw,Ci,Co]."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    n, h, wd, ci = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    oh, pt, pb = same_pads(h, kh, stride)
    ow, pl, pr = same_pads(wd, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, oh, ow, co), F64)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride, :]
            y += xs @ w[i, j]
    if b is not None:
        y += np.asarray(b, F64)  # tf.nn.bias_add, conv2d.py:216
    return y


def conv2d_same_grads(x, w, dy, stride=1):
    """Gradients of conv2d_same w.r.t. x, w, b (dgrad / wgrad / bias-grad)."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    dy = np.asarray(dy, F64)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    oh, pt, pb = same_pads(h, kh, stride)
    ow, pl, pr = same_pads(wd, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for i in range(kh):
        for j in range(kw):
            sl = (slice(None), slice(i, i + (oh - 1) * stride + 1, stride),
                  slice(j, j + (ow - 1) * stride + 1, stride), slice(None))
            dxp[sl] += dy @ w[i, j].T
            dw[i, j] = xp[sl].reshape(-1, ci).T @ dy.reshape(-1, co)
    dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    db = dy.sum(axis=(0, 1, 2))
    return dx, dw, db


def conv2d_direct_loops(x, w, b=None):
    """Pure-Python scalar loops, stride 1, SAME -- the slow independent check for tiny cases."""
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    _, pt, _ = same_pads(h, kh, 1)
    _, pl, _ = same_pads(wd, kw, 1)
    y = np.zeros((n, h, wd, co), F64)
    for nn in range(n):
        for oy in range(h):
            for ox in range(wd):
                for i in range(kh):
                    for j in range(kw):
                        iy, ix = oy + i - pt, ox + j - pl
                        if 0 <= iy < h and 0 <= ix < wd:
                            for c in range(ci):
                                for d in range(co):
                                    y[nn, oy, ox, d] += float(x[nn, iy, ix, c]) * float(w[i, j, c, d])
    if b is not None:
        y += b
    return y


# --------------------------------------------------------------------------------------
# tf.nn.conv2d_transpose, stride 2, SAME, filter [k,k,Cout,Cin]   (common/ops/deconv2d.py:99-109)
# --------------------------------------------------------------------------------------
def deconv2d_same(x, f, b=None, stride=2):
    """Transposed conv: the input-gradient of conv2d_same(stride) whose forward maps
    [N,sH,sW,Cout] -> [N,H,W,Cin] with filter f viewed as HWIO = [k,k,Cout,Cin]."""
    x = np.asarray(x, F64)
    f = np.asarray(f, F64)
    n, h, wd, cin = x.shape
    k, _, cout, cin2 = f.shape
    assert cin == cin2
    oh, ow = h * stride, wd * stride
    _, pt, pb = same_pads(oh, k, stride)
    _, pl, pr = same_pads(ow, k, stride)
    yp = np.zeros((n, oh + pt + pb, ow + pl + pr, cout), F64)
    for i in range(k):
        for j in range(k):
            yp[:, i:i + (h - 1) * stride + 1:stride, j:j + (wd - 1) * stride + 1:stride, :] += x @ f[i, j].T
    y = yp[:, pt:pt + oh, pl:pl + ow, :]
    if b is not None:
        y = y + np.asarray(b, F64)
    return y


def deconv2d_same_grads(x, f, dy, stride=2):
    """Gradients of deconv2d_same w.r.t. x, f, b."""
    x = np.asarray(x, F64)
    f = np.asarray(f, F64)
    dy = np.asarray(dy, F64)
    # deconv(x) = dgrad of conv; so d/dx deconv = conv forward of dy with f as HWIO [k,k,Cout,Cin]
    dx = conv2d_same(dy, f, None, stride)
    # df[i,j,co,ci] = sum dy_pad[n, s*p+i, s*q+j, co] * x[n,p,q,ci]  == wgrad of that conv with (input=dy, grad=x)
    _, df, _ = conv2d_same_grads(dy, f, x, stride)
    db = dy.sum(axis=(0, 1, 2))
    return dx, df, db


# --------------------------------------------------------------------------------------
# resampling helpers
# --------------------------------------------------------------------------------------
def upsample_nn2x(x):
    """tf.concat([x]*4, 3) + tf.depth_to_space(., 2)   (SNGAN/gan_cifar_resnet.py:143-145).
    depth_to_space uses DCR order: out[b,2h+i,2w+j,c] = in[b,h,w,(2i+j)*C+c]; on 4 identical
    copies this is an exact nearest-neighbour 2x upsample."""
    x = np.asarray(x)
    n, h, w, c = x.shape
    x4 = np.concatenate([x, x, x, x], axis=3)                       # [n,h,w,4c]
    y = x4.reshape(n, h, w, 2, 2, c).transpose(0, 1, 3, 2, 4, 5)     # [n,h,i,w,j,c]
    return y.reshape(n, 2 * h, 2 * w, c)


def upsample_nn2x_grad(dy):
    n, h2, w2, c = dy.shape
    return dy.reshape(n, h2 // 2, 2, w2 // 2, 2, c).sum(axis=(2, 4))


def meanpool2x2(x):
    """tf.add_n of the four stride-2 slices / 4.   (SNGAN/gan_cifar_resnet.py:120-121,129-130)"""
    return (x[:, ::2, ::2, :] + x[:, 1::2, ::2, :] + x[:, ::2, 1::2, :] + x[:, 1::2, 1::2, :]) / 4.


def meanpool2x2_grad(dy):
    return np.repeat(np.repeat(dy, 2, axis=1), 2, axis=2) / 4.


def relu(x):
    return np.maximum(x, 0.)  # tf.nn.relu, gan_cifar_resnet.py:82


def lrelu(x, leak=0.2):
    return np.maximum(x, leak * x)  # gan_cifar_resnet.py:85


# --------------------------------------------------------------------------------------
# spectral normalisation                                   (common/ops/sn.py:11-69)
# --------------------------------------------------------------------------------------
SN_EPS = 1e-12


def _l2normalize(v, eps=SN_EPS):
    return v / (np.sum(v ** 2) ** 0.5 + eps)  # sn.py:11-12 (eps OUTSIDE the sqrt)


def sn_forward(W, u, num_iters=1):
    """W any shape with Cout last; u [1,C].  Returns W_bar, u_final [1,C], sigma, cache.
    Literal restatement of sn.py:29-47,58-61 (power iteration, sigma = v W u^T, W/sigma)."""
    W = np.asarray(W, F64)
    shape = W.shape
    Wm = W.reshape(-1, shape[-1])                      # sn.py:30
    u_i = np.asarray(u, F64).reshape(1, -1)
    v_i = None
    for _ in range(num_iters):                          # sn.py:34-47
        v_i = _l2normalize(u_i @ Wm.T)                  # [1,K]
        u_i = _l2normalize(v_i @ Wm)                    # [1,C]
    sigma = (v_i @ Wm @ u_i.T)[0, 0]                    # sn.py:58
    W_bar = (Wm / sigma).reshape(shape)                 # sn.py:60-61
    return W_bar, u_i, sigma, v_i


def sn_backward(W, u, dW_bar):
    """Full gradient of one power-iteration step (NO stop_gradient anywhere in sn.py:34-61):
    a = W u, n=|a|, v = a/(n+e);  b = W^T v, m=|b|, u' = b/(m+e);  sigma = m^2/(m+e).
    d sigma/dW = s v b^T + g_a u^T,  s=(m+2e)/(m+e)^2,  g_v = s W b,
    g_a = g_v/(n+e) - a (a.g_v)/(n (n+e)^2);   dL/dW = G/sigma - (<G,W>/sigma^2) d sigma/dW."""
    W = np.asarray(W, F64)
    shape = W.shape
    Wm = W.reshape(-1, shape[-1])
    G = np.asarray(dW_bar, F64).reshape(Wm.shape)
    uu = np.asarray(u, F64).reshape(-1)
    e = SN_EPS
    a = Wm @ uu
    n = np.sqrt(np.sum(a * a))
    v = a / (n + e)
    b = Wm.T @ v
    m = np.sqrt(np.sum(b * b))
    sigma = m * m / (m + e)
    s = (m + 2 * e) / (m + e) ** 2
    g_v = s * (Wm @ b)
    g_a = g_v / (n + e) - a * (a @ g_v) / (n * (n + e) ** 2)
    dsig = s * np.outer(v, b) + np.outer(g_a, uu)
    GW = np.sum(G * Wm)
    dW = G / sigma - (GW / sigma ** 2) * dsig
    return dW.reshape(shape)


# --------------------------------------------------------------------------------------
# conditional batch norm                              (common/ops/normalization.py:27-59)
# --------------------------------------------------------------------------------------
BN_EPS = 1e-5


def cond_batchnorm_forward(x, labels, gamma, beta, groups=1, eps=BN_EPS):
    """Batch moments over (N,H,W) (tf.nn.moments: biased variance), per-sample gamma/beta rows
    gathered from [n_labels,C] tables, tf.nn.batch_normalization with eps=1e-5.
    `groups` > 1 splits the batch into that many consecutive towers with independent statistics
    (the reference builds one Generator call per tower, gan_cifar_resnet.py:326-332)."""
    x = np.asarray(x, F64)
    n = x.shape[0]
    assert n % groups == 0
    gs = n // groups
    xg = x.reshape(groups, gs, *x.shape[1:])
    mean = xg.mean(axis=(1, 2, 3), keepdims=True)
    var = ((xg - mean) ** 2).mean(axis=(1, 2, 3), keepdims=True)
    invstd = 1. / np.sqrt(var + eps)
    xhat = ((xg - mean) * invstd).reshape(x.shape)
    g = np.asarray(gamma, F64)[labels][:, None, None, :]
    bt = np.asarray(beta, F64)[labels][:, None, None, :]
    y = xhat * g + bt
    return y, (xhat, invstd, mean)


def cond_batchnorm_backward(dy, labels, gamma, cache, groups=1):
    xhat, invstd, _ = cache
    dy = np.asarray(dy, F64)
    gamma = np.asarray(gamma, F64)
    n = dy.shape[0]
    gs = n // groups
    dgamma = np.zeros_like(gamma)
    dbeta = np.zeros_like(gamma)
    np.add.at(dgamma, labels, (dy * xhat).sum(axis=(1, 2)))
    np.add.at(dbeta, labels, dy.sum(axis=(1, 2)))
    g = dy * gamma[labels][:, None, None, :]
    gg = g.reshape(groups, gs, *dy.shape[1:])
    xh = xhat.reshape(gg.shape)
    m1 = gg.mean(axis=(1, 2, 3), keepdims=True)
    m2 = (gg * xh).mean(axis=(1, 2, 3), keepdims=True)
    dx = (invstd * (gg - m1 - xh * m2)).reshape(dy.shape)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------------------
# linear / embedding / pooling / losses / optimiser
# --------------------------------------------------------------------------------------
def batch_norm_train(x, gamma, beta, moving, decay=0.9, eps=BN_EPS):
    """tf.contrib.layers.batch_norm(center, scale, is_training=True, fused=True, updates_collections=None,
    zero_debias_moving_mean=True)   (common/ops/normalization.py:8-24), one tower.
    Output: batch moments over (N,H,W), biased variance.  State (`moving` = dict moving_mean, moving_variance, biased,
    local_step; TF 1.5 moving_averages.assign_moving_average / _zero_debias):
        moving_variance <- mv - (1-decay)(mv - var_unbiased)     (fused batch norm reports the Bessel-corrected variance)
        biased <- biased - (1-decay)(biased - mean); local_step += 1; moving_mean <- biased / (1 - decay**local_step)
    Returns (y, cache for cond_batchnorm_backward, new_moving)."""
    x = np.asarray(x, F64)
    c = x.shape[-1]
    labels = np.zeros(x.shape[0], np.int64)
    y, cache = cond_batchnorm_forward(x, labels, np.asarray(gamma, F64).reshape(1, c), np.asarray(beta, F64).reshape(1, c), 1, eps)
    cnt = x.size // c
    mean = x.reshape(-1, c).mean(0)
    var_unbiased = x.reshape(-1, c).var(0) * cnt / max(cnt - 1, 1)
    new = dict(moving)
    new['moving_variance'] = moving['moving_variance'] - (1 - decay) * (moving['moving_variance'] - var_unbiased)
    new['biased'] = moving['biased'] - (1 - decay) * (moving['biased'] - mean)
    new['local_step'] = moving['local_step'] + 1
    new['moving_mean'] = new['biased'] / (1 - decay ** new['local_step'])
    return y, cache, new


def layer_norm_forward(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm, begin_norm_axis=1, begin_params_axis=-1 (common/ops/normalization.py:62-102):
    moments over all non-batch axes per sample (biased variance, variance_epsilon 1e-12), gamma/beta over the last axis."""
    x = np.asarray(x, F64)
    ax = tuple(range(1, x.ndim))
    mean = x.mean(axis=ax, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=ax, keepdims=True)
    invstd = 1. / np.sqrt(var + eps)
    xhat = (x - mean) * invstd
    return xhat * np.asarray(gamma, F64) + np.asarray(beta, F64), (xhat, invstd)


def layer_norm_backward(dy, gamma, cache):
    xhat, invstd = cache
    dy = np.asarray(dy, F64)
    ax = tuple(range(1, dy.ndim))
    red = tuple(range(dy.ndim - 1))
    g = dy * np.asarray(gamma, F64)
    dx = invstd * (g - g.mean(axis=ax, keepdims=True) - xhat * (g * xhat).mean(axis=ax, keepdims=True))
    return dx, (dy * xhat).sum(axis=red), dy.sum(axis=red)


def instance_norm_forward(x, gamma, beta, eps=1e-6):
    """tf.contrib.layers.instance_norm, NHWC (normalization.py:105-122): moments over (H,W) per sample and channel =
    conditional batch norm with one tower per sample and a one-row table."""
    x = np.asarray(x, F64)
    c = x.shape[-1]
    return cond_batchnorm_forward(x, np.zeros(x.shape[0], np.int64), np.asarray(gamma, F64).reshape(1, c),
                                  np.asarray(beta, F64).reshape(1, c), groups=x.shape[0], eps=eps)


def pixel_norm_forward(x, eps=1e-8):
    """normalization.py:125-140: x * rsqrt(mean(x*x, axis=3) + eps)"""
    x = np.asarray(x, F64)
    return x / np.sqrt((x * x).mean(axis=-1, keepdims=True) + eps)


def pixel_norm_backward(dy, x, eps=1e-8):
    x, dy = np.asarray(x, F64), np.asarray(dy, F64)
    a = 1. / np.sqrt((x * x).mean(axis=-1, keepdims=True) + eps)
    return a * dy - x * a ** 3 * (dy * x).mean(axis=-1, keepdims=True)


def linear(x, W, b=None):
    """tf.matmul + bias_add   (common/ops/linear.py:161-180)"""
    y = np.asarray(x, F64) @ np.asarray(W, F64)
    return y if b is None else y + np.asarray(b, F64)


def linear_grads(x, W, dy):
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


def embed_y(labels, table):
    """tf.nn.embedding_lookup   (common/ops/embedding.py:51)"""
    return np.asarray(table, F64)[labels]


def embed_y_grad(labels, dy, vocab):
    dt = np.zeros((vocab, dy.shape[1]), F64)
    np.add.at(dt, labels, dy)
    return dt


def relu_mean_hw(x):
    """nonlinearity + tf.reduce_mean(axis=[1,2])   (gan_cifar_resnet.py:299-301)"""
    return relu(x).mean(axis=(1, 2))


def hinge_d_loss(logits, n_real):
    """disc_real = logits[:n_real], disc_fake = logits[n_real:];
    mean(relu(1-real)) + mean(relu(1+fake))   (gan_cifar_resnet.py:362-363,379-381)."""
    logits = np.asarray(logits, F64)
    real, fake = logits[:n_real], logits[n_real:]
    loss = relu(1. - real).mean() + relu(1. + fake).mean()
    d = np.concatenate([-(1. - real > 0).astype(F64) / real.size, (1. + fake > 0).astype(F64) / fake.size])
    return loss, d


def hinge_g_loss(logits):
    """-mean(disc_fake)   (gan_cifar_resnet.py:492)"""
    logits = np.asarray(logits, F64)
    return -logits.mean(), np.full(logits.shape, -1. / logits.size, F64)


def softmax_xent(logits, labels):
    """tf.nn.sparse_softmax_cross_entropy_with_logits, mean over batch (gan_cifar_resnet.py:390-394)."""
    logits = np.asarray(logits, F64)
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    n = logits.shape[0]
    loss = -logp[np.arange(n), labels].mean()
    d = np.exp(logp)
    d[np.arange(n), labels] -= 1.
    return loss, d / n


def adam_tf_step(p, g, m, v, t, lr, beta1=0., beta2=0.9, eps=1e-8):
    """tf.train.AdamOptimizer (gan_cifar_resnet.py:521-526): lr_t = lr sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t m/(sqrt(v)+eps)  (eps outside the bias correction -- not the PyTorch formula)."""
    lr_t = lr * np.sqrt(1. - beta2 ** t) / (1. - beta1 ** t)
    m = beta1 * m + (1. - beta1) * g
    v = beta2 * v + (1. - beta2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


def lr_decay(iteration):
    """gan_cifar_resnet.py:454-457"""
    return max(0., 1. - iteration / 100000.) if iteration < 50000 else 0.5


def preprocess_real(data_u8, noise):
    """int -> float, /256, -.5, *2, + U[0,1/128), CHW-planar rows -> HWC rows
    (gan_cifar_resnet.py:334-337).  data_u8 [B,3072], noise [B,3072] in [0,1/128)."""
    x = 2. * (np.asarray(data_u8, F64) / 256. - .5) + np.asarray(noise, F64)
    b = x.shape[0]
    return x.reshape(b, 3, 32, 32).transpose(0, 2, 3, 1).reshape(b, 3072)
