"""torch-CPU float64 restatement of the PGGAN (Nvidia architecture) graph and train-step losses, with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned: the reference ships no tests or golden vectors for this
path and TensorFlow is not importable here).  Citations are relative to /root/reference.  Parameters live in a dict keyed by
the TF variable names the reference's scopes produce (`g_net/...`, `d_net/...`); the helpers shared with the SNGAN oracle
(SAME convolution, spectral norm with the full gradient, NN upsampling, 2x2 mean pool, the bf16-storage hook) come from
ref_torch.py.  Where PGGAN/model_nvidia.py does not run as written (a missing `pixelnorm` module, float channel counts, a
`reuse` argument Conv2D does not have) the intent stated in gan_lib_tensorflow_amd/PGGAN/model_nvidia.py is restated.
"""
import numpy as np
import torch

from . import ref_torch as T


def get_dim(stage):
    """model_nvidia.py:42-48"""
    return int(min(2048 // (2 ** stage), 512))


def lrelu(x, leak=0.2):
    return torch.maximum(x, leak * x)                                  # :15-17


def pixel_norm(x, eps=1e-8):
    """common/ops/normalization.py:125-140"""
    return x * torch.rsqrt((x * x).mean(dim=3, keepdim=True) + eps)


def minibatch_std(x):
    """model_nvidia.py:20-29"""
    m = x.mean(dim=0, keepdim=True)
    v = ((x - m) * (x - m)).mean(dim=0, keepdim=True)
    std = torch.sqrt(v + 1e-8).mean()
    return torch.cat([x, std.expand(x.shape[0], x.shape[1], x.shape[2], 1)], dim=3)


def minibatch_std_numpy(x):
    """the same statistic in plain NumPy (cross-check of the torch restatement)"""
    m = x.mean(axis=0, keepdims=True)
    v = ((x - m) ** 2).mean(axis=0, keepdims=True)
    std = np.sqrt(v + 1e-8).mean()
    return np.concatenate([x, np.full(x.shape[:3] + (1,), std, x.dtype)], axis=3)


def resize_bilinear(x, out_hw):
    """tf.image.resize_images(x, size) of TF 1.5 (method BILINEAR, align_corners=False): source coordinate = dst * in / out
    (no half-pixel offset), the two neighbours clamped to the last row / column.  x [N,H,W,C] NumPy or torch -> same type."""
    is_t = torch.is_tensor(x)
    a = x.detach().numpy() if is_t else np.asarray(x)
    n, hi, wi, c = a.shape
    ho, wo = out_hw
    ys, xs = np.arange(ho) * (hi / ho), np.arange(wo) * (wi / wo)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, hi - 1), np.minimum(x0 + 1, wi - 1)
    wy, wx = (ys - y0)[None, :, None, None], (xs - x0)[None, None, :, None]
    top = a[:, y0][:, :, x0] + (a[:, y0][:, :, x1] - a[:, y0][:, :, x0]) * wx
    bot = a[:, y1][:, :, x0] + (a[:, y1][:, :, x1] - a[:, y1][:, :, x0]) * wx
    out = top + (bot - top) * wy
    return torch.tensor(out) if is_t else out


# ------------------------------------------------------------------ parameters
def init_params(seed, bc, trans, z_dim=512):
    """Variables of model_nvidia.py for block_count bc, by name, with the initialisers of conv2d.py:83-140 / linear.py:76-80."""
    rng = np.random.default_rng(seed)
    P = {}

    def conv(scope, name, k, cin, cout, sn=False):
        P[f'{scope}/{name}/Filters'] = T.conv_init(rng, k, cin, cout, True)
        P[f'{scope}/{name}/Biases'] = np.zeros(cout, 'float32')
        if sn:
            P[f'{scope}/{name}/filters/spectral_norm/u'] = T.trunc_normal(rng, (1, cout))

    P['g_net/G.Input/W'] = T.linear_init(rng, z_dim, 4 * 4 * 512)
    P['g_net/G.Input/b'] = np.zeros(4 * 4 * 512, 'float32')
    conv('g_net', 'G.Conv', 3, 512, 512)
    c = 512
    for i in range(bc):
        d = get_dim(i)
        conv('g_net', f'G.UpBlock.{i + 1}/Conv.1', 3, c, d)
        conv('g_net', f'G.UpBlock.{i + 1}/Conv.2', 3, d, d)
        cprev, c = c, d
    if trans:
        conv('g_net', f'G.{bc}_toRGB1', 1, c, 3)
        conv('g_net', f'G.{bc}_toRGB2', 1, cprev, 3)
    else:
        conv('g_net', f'G.{bc}_toRGB', 1, c, 3)
    top = get_dim(bc - 1)
    if trans:
        conv('d_net', f'D.{bc}_fromRGB1', 1, 3, top, sn=True)
        conv('d_net', f'D.{bc}_fromRGB2', 1, 3, top, sn=True)
    else:
        conv('d_net', f'D.{bc}_fromRGB', 1, 3, top, sn=True)
    c = top
    for j in range(bc):               # D.Block.bc (out get_dim(bc-1)), then D.Block.(bc-i) (out get_dim(bc-1-i))
        name, out = f'D.Block.{bc - j}', get_dim(bc - 1 - j)
        conv('d_net', name + '/Conv.1', 3, c, c, sn=True)
        conv('d_net', name + '/Conv.2', 3, c, out, sn=True)
        c = out
    conv('d_net', 'D.Conv', 3, c + 1, top, sn=True)
    P['d_net/D.Output/W'] = T.linear_init(rng, top, 1)
    P['d_net/D.Output/b'] = np.zeros(1, 'float32')
    return P


# ------------------------------------------------------------------ model
class _Ctx(T._Ctx):
    """SN `u` write policy of the train step: update_u=True hands back u_final (update_collection=None); False leaves u"""

    def conv(self, x, name, sn=False):
        W = self.P[f'{self.scope}/{name}/Filters']
        if sn:
            key = f'{self.scope}/{name}/filters/spectral_norm/u'
            W, u_new, _ = T.spectral_normed_weight(W, self.P[key])
            if self.update_u:
                self.new_u[key] = u_new.detach()
        return T.conv2d_same(x, W, self.P[f'{self.scope}/{name}/Biases'])


def generator_block(c, x, name):
    h = T._st(c.conv(T.upsample_nn2x(x), name + '/Conv.1'))
    h = T._st(lrelu(T._st(pixel_norm(h))))
    h = T._st(c.conv(h, name + '/Conv.2'))
    return T._st(lrelu(T._st(pixel_norm(h))))


def generator(P, z, alpha, bc, trans):
    """model_nvidia.py:75-129 -> [N, 4 * 2**bc, 4 * 2**bc, 3]"""
    c = _Ctx(P, 'g_net', False)
    out = T._st(c.linear(z, 'G.Input')).reshape(-1, 4, 4, 512)
    out = T._st(lrelu(T._st(pixel_norm(out))))
    out = T._st(c.conv(out, 'G.Conv'))
    out = T._st(lrelu(T._st(pixel_norm(out))))
    for i in range(bc - 1):
        out = generator_block(c, out, f'G.UpBlock.{i + 1}')
    if trans:
        rgb1 = T._st(c.conv(generator_block(c, out, f'G.UpBlock.{bc}'), f'G.{bc}_toRGB1'))
        rgb2 = T._st(c.conv(T.upsample_nn2x(out), f'G.{bc}_toRGB2'))
        return T._st((1 - alpha) * rgb2 + alpha * rgb1)
    if bc > 0:
        out = generator_block(c, out, f'G.UpBlock.{bc}')
    return T._st(c.conv(out, f'G.{bc}_toRGB'))


def discriminator_block(c, x, name):
    h = T._st(lrelu(T._st(c.conv(x, name + '/Conv.1', sn=True))))
    h = T._st(lrelu(T._st(c.conv(h, name + '/Conv.2', sn=True))))
    return T._st(T.meanpool2x2(h))


def discriminator(P, x, alpha, bc, trans, update_u=False):
    """model_nvidia.py:164-237 -> (logits [N], {u name: u_final} when update_u)"""
    c = _Ctx(P, 'd_net', update_u)
    if trans:
        f1 = discriminator_block(c, T._st(c.conv(x, f'D.{bc}_fromRGB1', sn=True)), f'D.Block.{bc}')
        f2 = T._st(c.conv(T._st(T.meanpool2x2(x)), f'D.{bc}_fromRGB2', sn=True))
        h = T._st((1 - alpha) * f2 + alpha * f1)
    else:
        h = T._st(c.conv(x, f'D.{bc}_fromRGB', sn=True))
        if bc > 0:
            h = discriminator_block(c, h, f'D.Block.{bc}')
    for i in range(1, bc):
        h = discriminator_block(c, h, f'D.Block.{bc - i}')
    h = T._st(minibatch_std(h))
    h = T._st(lrelu(T._st(c.conv(h, 'D.Conv', sn=True))))
    h = T._st(h.mean(dim=(1, 2)))
    return T._st(c.linear(h, 'D.Output')).reshape(-1), c.new_u


def d_loss(P, real, z, alpha, bc, trans):
    """PGGAN/train.py:97-106: D(real) with update_collection=None (u advances), then D(G(z)) with NO_OPS reading the new u.
    Returns (loss, new_u)."""
    with torch.no_grad():
        x_fake = generator(P, z, alpha, bc, trans)
    disc_real, new_u = discriminator(P, real, alpha, bc, trans, update_u=True)
    P2 = dict(P)
    P2.update(new_u)
    disc_fake, _ = discriminator(P2, x_fake, alpha, bc, trans)
    return torch.relu(1. - disc_real).mean() + torch.relu(1. + disc_fake).mean(), new_u


def g_loss(P, z, alpha, bc, trans):
    """train.py:107"""
    disc_fake, _ = discriminator(P, generator(P, z, alpha, bc, trans), alpha, bc, trans)
    return -disc_fake.mean()
