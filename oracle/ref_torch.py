"""torch-CPU (float64 or float32) restatement of the SNGAN CIFAR-10 graph, with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).  This is the second,
independent implementation the NumPy oracle (`ref_ops.py`) is cross-checked against, the
network-level checker for the HIP path, and -- in float32 -- the `cpu_baseline` leg of bench.py
("CPU restatement of the reference graph", not TF1).

Citations are relative to /root/reference.  Parameters live in a plain dict keyed by the TF
variable names the reference's scopes produce (conv2d.py:59,142,170; sn.py:28,32;
linear.py:45,140,177; normalization.py:43,49-51; embedding.py:28,40).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

DIM_G = 128   # gan_cifar_resnet.py:41
DIM_D = 128   # gan_cifar_resnet.py:42
N_LABELS = 10
EMBEDDING_DIM = 300  # gan_cifar_resnet.py:60
SN_EPS = 1e-12
BN_EPS = 1e-5


# ------------------------------------------------------------------ initialisers
def _uniform(rng, stdev, size):
    return rng.uniform(-stdev * math.sqrt(3), stdev * math.sqrt(3), size=size).astype('float32')


def conv_init(rng, k, cin, cout, he_init=True, stride=1):
    """conv2d.py:83-140: fan_out divided by stride^2; he -> sqrt(4/(fi+fo)) else sqrt(2/(fi+fo))."""
    fan_in = cin * k * k
    fan_out = cout * k * k / (stride ** 2)
    std = math.sqrt(4. / (fan_in + fan_out)) if he_init else math.sqrt(2. / (fan_in + fan_out))
    return _uniform(rng, std, (k, k, cin, cout))


def linear_init(rng, cin, cout):
    """linear.py:76-80 (initialization=None falls in the glorot branch)."""
    return _uniform(rng, math.sqrt(2. / (cin + cout)), (cin, cout))


def trunc_normal(rng, size):
    """tf.truncated_normal_initializer() (sn.py:32): N(0,1) resampled outside 2 sigma."""
    out = rng.normal(size=size)
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = rng.normal(size=int(bad.sum()))
        bad = np.abs(out) > 2
    return out.astype('float32')


def init_sngan_params(seed=0):
    """All variables of Generator() + Discriminator() (gan_cifar_resnet.py:237-313), creation order."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()

    def conv(scope, name, k, cin, cout, he_init=True, sn=False):
        P[f'{scope}/{name}/Filters'] = conv_init(rng, k, cin, cout, he_init)
        if sn:
            P[f'{scope}/{name}/filters/spectral_norm/u'] = trunc_normal(rng, (1, cout))
        P[f'{scope}/{name}/Biases'] = np.zeros(cout, 'float32')

    def lin(scope, name, cin, cout, sn=False):
        P[f'{scope}/{name}/W'] = linear_init(rng, cin, cout)
        if sn:
            P[f'{scope}/{name}/spectral_norm/u'] = trunc_normal(rng, (1, cout))
        P[f'{scope}/{name}/b'] = np.zeros(cout, 'float32')

    def cbn(scope, name, c):
        P[f'{scope}/{name}/CondBatchNorm/offset'] = np.zeros((N_LABELS, c), 'float32')
        P[f'{scope}/{name}/CondBatchNorm/scale'] = np.ones((N_LABELS, c), 'float32')

    g = 'Generator'
    lin(g, 'G.Input', 128, 4 * 4 * DIM_G * 8)
    for i, (ci, co) in enumerate([(DIM_G * 8, DIM_G * 2), (DIM_G * 2, DIM_G * 2), (DIM_G * 2, DIM_G * 2)], 1):
        conv(g, f'G.Block.{i}.Shortcut', 1, ci, co, he_init=False)
        cbn(g, f'G.Block.{i}.N1', ci)
        conv(g, f'G.Block.{i}.Conv1', 3, ci, co)
        cbn(g, f'G.Block.{i}.N2', co)
        conv(g, f'G.Block.{i}.Conv2', 3, co, co)
    cbn(g, 'G.OutputNorm', DIM_G * 2)
    conv(g, 'G.Output', 3, DIM_G * 2, 3, he_init=False)

    d = 'Discriminator'
    conv(d, 'D.Block.1.Shortcut', 1, 3, DIM_D, he_init=False, sn=True)
    conv(d, 'D.Block.1.Conv1', 3, 3, DIM_D, sn=True)
    conv(d, 'D.Block.1.Conv2', 3, DIM_D, DIM_D, sn=True)
    P[f'{d}/Embedding.Label/embedding_map'] = rng.uniform(-0.08, 0.08, (N_LABELS, EMBEDDING_DIM)).astype('float32')
    lin(d, 'D.Embedding_y', EMBEDDING_DIM, DIM_D, sn=True)
    conv(d, 'D.Block.2.Shortcut', 1, DIM_D * 2, DIM_D, he_init=False, sn=True)
    conv(d, 'D.Block.2.Conv1', 3, DIM_D * 2, DIM_D * 2, sn=True)
    conv(d, 'D.Block.2.Conv2', 3, DIM_D * 2, DIM_D, sn=True)
    for i in (3, 4):
        conv(d, f'D.Block.{i}.Conv1', 3, DIM_D, DIM_D, sn=True)
        conv(d, f'D.Block.{i}.Conv2', 3, DIM_D, DIM_D, sn=True)
    lin(d, 'D.Output', DIM_D, 1, sn=True)
    return P


def is_state(name):
    """Non-trainable variables: the SN `u` vectors (sn.py:32 trainable=False) and batch-norm moving statistics."""
    return name.endswith(('spectral_norm/u', '/moving_mean', '/moving_variance', '/moving_mean/biased', '/moving_mean/local_step'))


def to_torch(P, dtype=torch.float64, requires_grad=True):
    out = OrderedDict()
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad and not is_state(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ------------------------------------------------------------------ ops
def conv2d_same(x, w, b=None):
    """NHWC/HWIO cross-correlation, SAME, stride 1 (conv2d.py:180-187,212-216)."""
    k = w.shape[0]
    pt = (k - 1) // 2
    pb = (k - 1) - pt
    xn = x.permute(0, 3, 1, 2)
    if k > 1:
        xn = F.pad(xn, (pt, pb, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1))
    y = y.permute(0, 2, 3, 1)
    return y if b is None else y + b


def l2normalize(v, eps=SN_EPS):
    return v / (torch.sum(v ** 2) ** 0.5 + eps)  # sn.py:11-12


def spectral_normed_weight(W, u, num_iters=1):
    """sn.py:29-61, differentiable through the power iteration (no stop_gradient)."""
    Wm = W.reshape(-1, W.shape[-1])
    u_i, v_i = u, None
    for _ in range(num_iters):
        v_i = l2normalize(u_i @ Wm.t())
        u_i = l2normalize(v_i @ Wm)
    sigma = (v_i @ Wm @ u_i.t())[0, 0]
    return (Wm / sigma).reshape(W.shape), u_i, sigma


def upsample_nn2x(x):
    """concat x4 + depth_to_space(2) == exact NN upsample (gan_cifar_resnet.py:143-145)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def meanpool2x2(x):
    return (x[:, ::2, ::2, :] + x[:, 1::2, ::2, :] + x[:, ::2, 1::2, :] + x[:, 1::2, 1::2, :]) / 4.


def cond_batchnorm(x, labels, scale, offset, groups=1):
    """normalization.py:47-57; `groups` = number of towers with independent batch statistics."""
    n = x.shape[0]
    xg = x.reshape(groups, n // groups, *x.shape[1:])
    mean = xg.mean(dim=(1, 2, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    xhat = ((xg - mean) * torch.rsqrt(var + BN_EPS)).reshape(x.shape)
    return xhat * scale[labels][:, None, None, :] + offset[labels][:, None, None, :]


# ------------------------------------------------------------------ storage emulation (test instrument)
# The product path keeps every activation and activation gradient in bf16 between kernels.  STORE, when set, is applied
# to every tensor that path materialises (conv / linear outputs, normalised+activated tensors, block outputs), so a test
# can measure what bf16 STORAGE alone does to losses and gradients, independently of any kernel: see
# tests/test_oracle.py::test_bf16_storage_sensitivity_of_generator_gradients.  None (the default) = exact arithmetic.
STORE = None


class _RoundBoth(torch.autograd.Function):
    """value and gradient both rounded to bf16 (what a bf16 tensor in HBM does to the forward and backward streams)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """value rounded, gradient exact"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """value exact, gradient rounded"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def bf16_storage(x):
    return _RoundBoth.apply(x)


def bf16_storage_fwd(x):
    return _RoundFwd.apply(x)


def bf16_storage_bwd(x):
    return _RoundBwd.apply(x)


# Attribution instrument: every stored tensor carries a class tag -- 'lin' (G.Input output), 'short' (shortcut conv outputs),
# 'cbn' (normalised + activated tensors), 'conv1' (conv_1 outputs = the next norm's input), 'block' (block outputs = shortcut +
# conv_2), 'image' (tanh output), 'd' (everything the critic stores).  Classes listed in STORE_EXACT are left unrounded, so a test
# or script can un-round ONE class at a time and see which stored tensor carries the gradient error.
STORE_EXACT = frozenset()


def _st(x, cls=None):
    if STORE is None or (cls is not None and cls in STORE_EXACT):
        return x
    return STORE(x)


# ------------------------------------------------------------------ model
class _Ctx:
    """Mimics the reference's variable scopes: fetches variables by name, records new SN u."""

    def __init__(self, P, scope, update_u):
        self.P, self.scope, self.update_u = P, scope, update_u
        self.new_u = {}

    def conv(self, x, name, sn=False):
        W = self.P[f'{self.scope}/{name}/Filters']
        if sn:
            key = f'{self.scope}/{name}/filters/spectral_norm/u'
            W, u_new, _ = spectral_normed_weight(W, self.P[key])
            self.new_u[key] = u_new.detach()
        return conv2d_same(x, W, self.P[f'{self.scope}/{name}/Biases'])          # stored by the caller (after the fused epilogue)

    def linear(self, x, name, sn=False):
        W = self.P[f'{self.scope}/{name}/W']
        if sn:
            key = f'{self.scope}/{name}/spectral_norm/u'
            W, u_new, _ = spectral_normed_weight(W, self.P[key])
            self.new_u[key] = u_new.detach()
        return x @ W + self.P[f'{self.scope}/{name}/b']

    def cbn(self, x, name, labels, groups):
        return cond_batchnorm(x, labels, self.P[f'{self.scope}/{name}/CondBatchNorm/scale'],
                              self.P[f'{self.scope}/{name}/CondBatchNorm/offset'], groups)


def generator(P, noise, labels, groups=1):
    """gan_cifar_resnet.py:237-263.  noise [n,128] -> [n,3072] (HWC order, tanh range)."""
    c = _Ctx(P, 'Generator', False)
    out = _st(c.linear(noise, 'G.Input'), 'lin').reshape(-1, 4, 4, DIM_G * 8)
    for i in (1, 2, 3):
        name = f'G.Block.{i}'
        shortcut = _st(c.conv(upsample_nn2x(out), name + '.Shortcut'), 'short')     # :179-182 -> :140-151
        h = _st(torch.relu(c.cbn(out, name + '.N1', labels, groups)), 'cbn')
        h = _st(c.conv(upsample_nn2x(h), name + '.Conv1'), 'conv1')                 # :192-195
        h = _st(torch.relu(c.cbn(h, name + '.N2', labels, groups)), 'cbn')
        h = c.conv(h, name + '.Conv2')
        out = _st(shortcut + h, 'block')
    out = _st(torch.relu(c.cbn(out, 'G.OutputNorm', labels, groups)), 'cbn')
    out = _st(torch.tanh(c.conv(out, 'G.Output')), 'image')
    return out.reshape(-1, 3072)


def discriminator(P, x, labels):
    """gan_cifar_resnet.py:266-313 (ACGAN=False).  Returns logits [n] and {u name: new u}."""
    c = _Ctx(P, 'Discriminator', True)
    x = x.reshape(-1, 32, 32, 3)
    shortcut = _st(c.conv(meanpool2x2(x), 'D.Block.1.Shortcut', sn=True), 'd')  # :218-221 -> :125-135
    h = _st(c.conv(x, 'D.Block.1.Conv1', sn=True), 'd')
    h = meanpool2x2(c.conv(torch.relu(h), 'D.Block.1.Conv2', sn=True))
    out = _st(shortcut + h, 'd')
    emb = _st(c.linear(P['Discriminator/Embedding.Label/embedding_map'][labels], 'D.Embedding_y', sn=True), 'd')
    emb = emb[:, None, None, :].expand(-1, out.shape[1], out.shape[2], -1)
    out = torch.cat([out, emb], dim=3)                                      # :282-284
    # D.Block.2, resample='down'
    shortcut = _st(meanpool2x2(c.conv(out, 'D.Block.2.Shortcut', sn=True)), 'd')    # ConvMeanPool :112-122
    h = _st(c.conv(torch.relu(out), 'D.Block.2.Conv1', sn=True), 'd')
    h = meanpool2x2(c.conv(torch.relu(h), 'D.Block.2.Conv2', sn=True))
    out = _st(shortcut + h, 'd')
    for i in (3, 4):                                                        # identity shortcut :176-177
        h = _st(c.conv(torch.relu(out), f'D.Block.{i}.Conv1', sn=True), 'd')
        h = c.conv(torch.relu(h), f'D.Block.{i}.Conv2', sn=True)
        out = _st(out + h, 'd')
    out = _st(torch.relu(out).mean(dim=(1, 2)), 'd')                             # :299-301
    logits = _st(c.linear(out, 'D.Output', sn=True), 'd').reshape(-1)
    return logits, c.new_u


def preprocess_real(data_u8, deq_noise, dtype):
    """gan_cifar_resnet.py:334-337"""
    x = 2. * (data_u8.to(dtype) / 256. - .5) + deq_noise.to(dtype)
    b = x.shape[0]
    return x.reshape(b, 3, 32, 32).permute(0, 2, 3, 1).reshape(b, 3072)


def d_loss_fn(P, real_u8, labels, z, deq_noise, towers=2, real_pre=None):
    """One D-step loss (gan_cifar_resnet.py:326-381): fakes from `towers` Generator calls of
    B/towers samples each, conditioned on the REAL labels; D on concat(real, fake).
    `real_pre` (already preprocessed [B,3072]) overrides real_u8/deq_noise when given."""
    dtype = z.dtype
    fake = generator(P, z, labels, groups=towers)
    real = preprocess_real(real_u8, deq_noise, dtype) if real_pre is None else real_pre
    logits, new_u = discriminator(P, torch.cat([real, fake], 0), torch.cat([labels, labels], 0))
    b = real.shape[0]
    loss = torch.relu(1. - logits[:b]).mean() + torch.relu(1. + logits[b:]).mean()
    return loss, new_u, logits


def g_loss_fn(P, z, fake_labels, towers=2):
    """One G-step loss (gan_cifar_resnet.py:464-498): per tower -mean(D(G(z))); towers averaged.
    D is called with update_collection=NO_OPS: u is read, never written (sn.py:62-65)."""
    fake = generator(P, z, fake_labels, groups=towers)
    logits, _ = discriminator(P, fake, fake_labels)
    return -logits.mean(), logits


def sngan_losses(logits_d, b, logits_g=None, loss_type='HINGE', soft_plus=False):
    """The LOSS_TYPE / SOFT_PLUS switch of gan_cifar_resnet.py:364-386 (critic: logits_d = concat(real [b], fake)) and :483-497
    (generator: logits_g), expression for expression.  -> (disc_cost | None, gen_cost | None)"""
    import torch.nn.functional as F
    sp = F.softplus
    d = g = None
    if logits_d is not None:
        real, fake = logits_d[:b], logits_d[b:]
        if loss_type == 'Goodfellow':
            if soft_plus:
                d = -sp(torch.log(torch.sigmoid(real))).mean() - sp(torch.log(1 - torch.sigmoid(fake))).mean()
            else:
                d = -torch.log(torch.sigmoid(real)).mean() - torch.log(1 - torch.sigmoid(fake)).mean()
        elif loss_type == 'HINGE':
            if soft_plus:
                zero = torch.zeros((), dtype=real.dtype)
                d = sp(-torch.minimum(zero, -1 + real)).mean() + sp(-torch.minimum(zero, -1 - fake)).mean()
            else:
                d = torch.relu(1. - real).mean() + torch.relu(1. + fake).mean()
        elif loss_type == 'WGAN':
            d = (sp(fake).mean() + sp(-real).mean()) if soft_plus else (fake.mean() - real.mean())
        else:
            raise ValueError(loss_type)
    if logits_g is not None:
        if loss_type == 'Goodfellow':
            g = sp(-torch.log(torch.sigmoid(logits_g))).mean() if soft_plus else -torch.log(torch.sigmoid(logits_g)).mean()
        else:
            g = sp(-logits_g).mean() if soft_plus else -logits_g.mean()
    return d, g


def lr_decay(iteration):
    return max(0., 1. - iteration / 100000.) if iteration < 50000 else 0.5


class AdamTF:
    """tf.train.AdamOptimizer(beta1=0, beta2=0.9) (gan_cifar_resnet.py:521-526)."""

    def __init__(self, names, beta1=0., beta2=0.9, eps=1e-8):
        self.names, self.b1, self.b2, self.eps, self.t = list(names), beta1, beta2, eps, 0
        self.m, self.v = {}, {}

    @torch.no_grad()
    def step(self, P, grads, lr):
        self.t += 1
        lr_t = lr * math.sqrt(1. - self.b2 ** self.t) / (1. - self.b1 ** self.t)
        for k, g in zip(self.names, grads):
            m = self.m.get(k, torch.zeros_like(g))
            v = self.v.get(k, torch.zeros_like(g))
            m = self.b1 * m + (1 - self.b1) * g
            v = self.b2 * v + (1 - self.b2) * g * g
            self.m[k], self.v[k] = m, v
            P[k] -= lr_t * m / (v.sqrt() + self.eps)


def trainable_names(P, scope):
    return [k for k in P if k.startswith(scope + '/') and not is_state(k)]


class Trainer:
    """The reference's loop body (gan_cifar_resnet.py:599-620) on explicit inputs."""

    def __init__(self, P, lr=2e-4):
        self.P, self.lr = P, lr
        self.g_names = trainable_names(P, 'Generator')
        self.d_names = trainable_names(P, 'Discriminator')
        self.g_opt, self.d_opt = AdamTF(self.g_names), AdamTF(self.d_names)

    def d_step(self, iteration, real_u8, labels, z, deq_noise, real_pre=None):
        loss, new_u, _ = d_loss_fn(self.P, real_u8, labels, z, deq_noise, real_pre=real_pre)
        grads = torch.autograd.grad(loss, [self.P[k] for k in self.d_names])
        with torch.no_grad():
            for k, u in new_u.items():          # update_collection=None: u <- u_final (sn.py:55-56)
                self.P[k].copy_(u)
        self.d_opt.step(self.P, grads, self.lr * lr_decay(iteration))
        return float(loss)

    def g_step(self, iteration, z, fake_labels):
        loss, _ = g_loss_fn(self.P, z, fake_labels)
        grads = torch.autograd.grad(loss, [self.P[k] for k in self.g_names])
        self.g_opt.step(self.P, grads, self.lr * lr_decay(iteration))
        return float(loss)


# ================================================================== ACGAN configuration (BASELINE.json config 3)
def init_acgan_params(seed=0, z_dim=128):
    """All variables of ACGAN.get_generator / get_discriminator (ACGAN/model.py:21-90), scopes g_net / d_net."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()

    def conv(scope, name, k, cin, cout, he_init=True):
        P[f'{scope}/{name}/Filters'] = conv_init(rng, k, cin, cout, he_init)
        P[f'{scope}/{name}/Biases'] = np.zeros(cout, 'float32')

    def lin(scope, name, cin, cout):
        P[f'{scope}/{name}/W'] = linear_init(rng, cin, cout)
        P[f'{scope}/{name}/b'] = np.zeros(cout, 'float32')

    def cbn(scope, name, c):
        P[f'{scope}/{name}/CondBatchNorm/offset'] = np.zeros((N_LABELS, c), 'float32')
        P[f'{scope}/{name}/CondBatchNorm/scale'] = np.ones((N_LABELS, c), 'float32')

    def bn(scope, name, c):
        P[f'{scope}/{name}/BatchNorm/beta'] = np.zeros((1, c), 'float32')
        P[f'{scope}/{name}/BatchNorm/gamma'] = np.ones((1, c), 'float32')
        P[f'{scope}/{name}/BatchNorm/moving_mean'] = np.zeros(c, 'float32')
        P[f'{scope}/{name}/BatchNorm/moving_variance'] = np.ones(c, 'float32')
        P[f'{scope}/{name}/BatchNorm/moving_mean/biased'] = np.zeros(c, 'float32')
        P[f'{scope}/{name}/BatchNorm/moving_mean/local_step'] = np.zeros(1, 'float32')

    g = 'g_net'
    lin(g, 'G.Input', z_dim, 4 * 4 * 1024)
    for i, (ci, co) in enumerate([(1024, 256), (256, 256), (256, 256)], 1):
        conv(g, f'G.{i}.Shortcut', 1, ci, co, he_init=False)
        cbn(g, f'G.{i}.N1', ci)
        conv(g, f'G.{i}.Conv1', 3, ci, co)
        cbn(g, f'G.{i}.N2', co)
        conv(g, f'G.{i}.Conv2', 3, co, co)
    bn(g, 'G.OutputN', 256)
    conv(g, 'G.Output', 3, 256, 3, he_init=False)
    d = 'd_net'
    conv(d, 'D.DownBlock.1.Shortcut', 1, 3, 128, he_init=False)
    conv(d, 'D.DownBlock.1.Conv1', 3, 3, 128)
    conv(d, 'D.DownBlock.1.Conv2', 3, 128, 128)
    for name, down in (('D.DownBlock.2', True), ('D.NoneBlock.3', False), ('D.NoneBlock.4', False)):
        if down:
            conv(d, name + '.Shortcut', 1, 128, 128, he_init=False)
        bn(d, name + '.N1', 128)
        conv(d, name + '.Conv1', 3, 128, 128)
        bn(d, name + '.N2', 128)
        conv(d, name + '.Conv2', 3, 128, 128)
    lin(d, 'D.Output', 128, 1)
    lin(d, 'D.ACGANOutput', 128, 10)
    return P


def batch_norm_train(x, gamma, beta, eps=BN_EPS):
    """tf.contrib.layers.batch_norm(is_training=True): biased batch variance over (N, H, W)  (normalization.py:8-24)"""
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.reshape(1, 1, 1, -1) + beta.reshape(1, 1, 1, -1)


def lrelu(x, leak=0.2):
    return torch.maximum(x, leak * x)


def acgan_generator(P, z, labels):
    """ACGAN/model.py:31-47 -> [N, 32, 32, 3]"""
    c = _Ctx(P, 'g_net', False)
    out = _st(c.linear(z, 'G.Input')).reshape(-1, 4, 4, 1024)
    for i in (1, 2, 3):
        name = f'G.{i}'
        shortcut = _st(c.conv(upsample_nn2x(out), name + '.Shortcut'))
        h = _st(torch.relu(c.cbn(out, name + '.N1', labels, 1)))
        h = _st(c.conv(upsample_nn2x(h), name + '.Conv1'))
        h = _st(torch.relu(c.cbn(h, name + '.N2', labels, 1)))
        out = _st(shortcut + c.conv(h, name + '.Conv2'))
    out = _st(torch.relu(batch_norm_train(out, P['g_net/G.OutputN/BatchNorm/gamma'], P['g_net/G.OutputN/BatchNorm/beta'])))
    return _st(torch.tanh(c.conv(out, 'G.Output')))


def acgan_discriminator(P, x):
    """ACGAN/model.py:59-88: x [N, 32, 32, 3] -> (logits [N], class logits [N, 10]); batch norm uses batch statistics"""
    c = _Ctx(P, 'd_net', False)

    def bn(h, name):
        return batch_norm_train(h, P[f'd_net/{name}/BatchNorm/gamma'], P[f'd_net/{name}/BatchNorm/beta'])
    shortcut = _st(c.conv(meanpool2x2(x), 'D.DownBlock.1.Shortcut'))
    h = _st(c.conv(x, 'D.DownBlock.1.Conv1'))
    h = _st(lrelu(h))
    h = _st(meanpool2x2(_st(c.conv(h, 'D.DownBlock.1.Conv2'))))
    out = _st(shortcut + h)
    for name, down in (('D.DownBlock.2', True), ('D.NoneBlock.3', False), ('D.NoneBlock.4', False)):
        sc = _st(meanpool2x2(_st(c.conv(out, name + '.Shortcut')))) if down else out
        h = _st(lrelu(_st(bn(out, name + '.N1'))))
        h = _st(c.conv(h, name + '.Conv1'))
        h = _st(lrelu(_st(bn(h, name + '.N2'))))
        h = _st(c.conv(h, name + '.Conv2'))
        if down:
            h = _st(meanpool2x2(h))
        out = _st(sc + h)
    out = _st(_st(lrelu(out)).mean(dim=(1, 2)))
    return _st(c.linear(out, 'D.Output')).reshape(-1), _st(c.linear(out, 'D.ACGANOutput'))


def acgan_d_loss(P, real, real_labels, z, fake_labels, alpha):
    """ACGAN/train.py:89-115: hinge + 10 * gradient penalty + class cross-entropy on the real batch.
    real [N,32,32,3] (preprocessed), alpha [N].  Returns (total, parts dict)."""
    with torch.no_grad():
        x_fake = acgan_generator(P, z, fake_labels)
    disc_real, ac_real = acgan_discriminator(P, real)
    disc_fake, _ = acgan_discriminator(P, x_fake)
    d_gan = torch.relu(1. - disc_real).mean() + torch.relu(1. + disc_fake).mean()
    interp = (real + alpha.reshape(-1, 1, 1, 1) * (x_fake - real)).detach().requires_grad_(True)
    d_int, _ = acgan_discriminator(P, interp)
    (grads,) = torch.autograd.grad(d_int.sum(), interp, create_graph=True)
    slopes = torch.sqrt((grads ** 2).sum(dim=(1, 2, 3)) + 1e-10)
    gp = 10. * ((slopes - 1.) ** 2).mean()
    d_ac = F.cross_entropy(ac_real, real_labels.long())
    return d_gan + gp + d_ac, dict(d_gan=d_gan, gp=gp, d_ac=d_ac, disc_real=disc_real, disc_fake=disc_fake, grads=grads, x_fake=x_fake)


def acgan_g_loss(P, z, fake_labels, acgan_scale_G=0.1):
    """ACGAN/train.py:117-121"""
    x_fake = acgan_generator(P, z, fake_labels)
    disc_fake, ac_fake = acgan_discriminator(P, x_fake)
    g_gan = -disc_fake.mean()
    g_ac = F.cross_entropy(ac_fake, fake_labels.long())
    return g_gan + acgan_scale_G * g_ac, dict(g_gan=g_gan, g_ac=g_ac)
