"""torch-CPU float64 restatement of the Pix2Pix U-Net / PatchGAN graph and losses, with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned: the reference ships no tests or golden vectors for this
path and TensorFlow is not importable here).  Citations are relative to /root/reference.  Parameters live in a dict keyed by
the TF variable names (`g_net/encoder_1/Conv2D/Filters`, `g_net/encoder_2/InstanceNorm/beta`, `d_net/layer_1/Conv2D/filters/
spectral_norm/u`, ...).  Dropout masks are explicit inputs (tf.nn.dropout draws them; a parity run must share them).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ref_torch as T


def conv2d_tf(x, w, b=None, stride=1, padding='SAME', pad_input=0):
    """tf.nn.conv2d on NHWC / HWIO.  SAME: out = ceil(in / stride), total pad = max((out-1)*stride + k - in, 0), the smaller
    half in front (conv2d.py:180-187); VALID after tf.pad(pad_input) (Pix2Pix/networks.py:482-484)."""
    k = w.shape[0]
    n, h, wd, c = x.shape
    xn = x.permute(0, 3, 1, 2)
    if padding == 'SAME':
        oh, ow = -(-h // stride), -(-wd // stride)
        ph, pw = max((oh - 1) * stride + k - h, 0), max((ow - 1) * stride + k - wd, 0)
        xn = F.pad(xn, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    elif pad_input:
        xn = F.pad(xn, (pad_input,) * 4)
    y = F.conv2d(xn, w.permute(3, 2, 0, 1).contiguous(), stride=stride).permute(0, 2, 3, 1)
    return y if b is None else y + b


def conv2d_numpy(x, w, b, stride, pad, out_hw):
    """the general gather in plain loops (small cases): y[n,oy,ox,co] = sum x[n, oy*s + a - pad, ox*s + b - pad, ci] w[a,b,ci,co]"""
    n, h, wd, cin = x.shape
    k, _, _, cout = w.shape
    y = np.zeros((n, out_hw[0], out_hw[1], cout))
    for oy in range(out_hw[0]):
        for ox in range(out_hw[1]):
            for a in range(k):
                for bb in range(k):
                    iy, ix = oy * stride + a - pad, ox * stride + bb - pad
                    if 0 <= iy < h and 0 <= ix < wd:
                        y[:, oy, ox, :] += x[:, iy, ix, :] @ w[a, bb]
    return y + (0 if b is None else b)


def instance_norm(x, gamma, beta, eps=1e-5):
    """tf.contrib.layers.instance_norm(center, scale): moments over (H, W) per sample and channel, biased variance
    (common/ops/normalization.py:105-122; epsilon 1e-5 from norm_layer, Pix2Pix/networks.py:25-43)"""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.reshape(1, 1, 1, -1) + beta.reshape(1, 1, 1, -1)


def lrelu(x, leak=0.2):
    return torch.maximum(x, leak * x)


ENC = lambda ngf: [ngf, ngf * 2, ngf * 4] + [ngf * 8] * 6                          # noqa: E731  (networks.py:364-379)
DEC = lambda ngf: [(ngf * 8, 0.5)] * 3 + [(ngf * 8, 0.0)] * 2 + [(ngf * 4, 0.0), (ngf * 2, 0.0), (ngf, 0.0)]   # noqa: E731  (:400-409)


def init_params(seed, ngf=64, ndf=64, in_ch=3, out_ch=3):
    rng = np.random.default_rng(seed)
    P = {}

    def conv(scope, cin, cout, stride, sn=False):
        P[f'{scope}/Conv2D/Filters'] = T.conv_init(rng, 4, cin, cout, True, stride)
        P[f'{scope}/Conv2D/Biases'] = np.zeros(cout, 'float32')
        if sn:
            P[f'{scope}/Conv2D/filters/spectral_norm/u'] = T.trunc_normal(rng, (1, cout))

    def inorm(scope, c):
        P[f'{scope}/InstanceNorm/beta'] = np.zeros((1, c), 'float32')
        P[f'{scope}/InstanceNorm/gamma'] = np.ones((1, c), 'float32')

    enc = ENC(ngf)
    c = in_ch
    for i, co in enumerate(enc):
        conv(f'g_net/encoder_{i + 1}', c, co, 2)
        if i > 0:
            inorm(f'g_net/encoder_{i + 1}', co)
        c = co
    for j, (co, _) in enumerate(DEC(ngf)):
        skip = len(enc) - j - 1
        cin = c if j == 0 else c + enc[skip]
        conv(f'g_net/decoder_{skip + 1}', cin, co, 1)
        inorm(f'g_net/decoder_{skip + 1}', co)
        c = co
    conv('g_net/decoder_1', c + enc[0], out_ch, 1)
    c = in_ch + out_ch
    for i, (co, st) in enumerate([(ndf, 2), (ndf * 2, 2), (ndf * 4, 2), (ndf * 8, 2), (ndf * 8, 1), (1, 1)]):
        conv(f'd_net/layer_{i + 1}', c, co, st, sn=True)
        c = co
    return P


def generator(P, x, masks, ngf=64):
    """networks.py:359-468.  masks: {decoder scope name: 0/1 tensor like that decoder's output} (keep_prob 0.5)"""
    enc = ENC(ngf)
    layers = []
    h = x
    for i in range(len(enc)):
        s = f'g_net/encoder_{i + 1}'
        if i > 0:
            h = T._st(lrelu(h))
        h = T._st(conv2d_tf(h, P[s + '/Conv2D/Filters'], P[s + '/Conv2D/Biases'], 2, 'SAME'))
        if i > 0:
            h = T._st(instance_norm(h, P[s + '/InstanceNorm/gamma'], P[s + '/InstanceNorm/beta']))
        layers.append(h)
    for j, (_, drop) in enumerate(DEC(ngf)):
        skip = len(enc) - j - 1
        s = f'g_net/decoder_{skip + 1}'
        inp = layers[-1] if j == 0 else torch.cat([layers[-1], layers[skip]], dim=3)
        h = T._st(conv2d_tf(T.upsample_nn2x(torch.relu(inp)), P[s + '/Conv2D/Filters'], P[s + '/Conv2D/Biases'], 1, 'SAME'))
        h = T._st(instance_norm(h, P[s + '/InstanceNorm/gamma'], P[s + '/InstanceNorm/beta']))
        if drop > 0.0:
            h = T._st(h * masks[s] / (1.0 - drop))
        layers.append(h)
    inp = torch.cat([layers[-1], layers[0]], dim=3)
    s = 'g_net/decoder_1'
    return T._st(torch.tanh(conv2d_tf(T.upsample_nn2x(torch.relu(inp)), P[s + '/Conv2D/Filters'], P[s + '/Conv2D/Biases'], 1, 'SAME')))


def discriminator(P, a, b):
    """networks.py:471-536 with spectral norm, update_collection=None.  -> (patch logits [N,h,w,1], {u name: u_final})"""
    h = torch.cat([a, b], dim=3)
    new_u = {}
    for i, st in enumerate([2, 2, 2, 2, 1, 1]):
        s = f'd_net/layer_{i + 1}/Conv2D'
        W, u_new, _ = T.spectral_normed_weight(P[s + '/Filters'], P[s + '/filters/spectral_norm/u'])
        new_u[s + '/filters/spectral_norm/u'] = u_new.detach()
        h = T._st(conv2d_tf(h, W, P[s + '/Biases'], st, 'VALID', pad_input=1))
        if i < 5:
            h = T._st(lrelu(h))
    return h, new_u


def d_loss(P, a, b, masks, ngf=64):
    """train.py:452-483 (HINGE): real pass, then the fake pass on the u the real pass wrote.  -> (loss, u after both passes)"""
    with torch.no_grad():
        out = generator(P, a, masks, ngf)
    pr, u1 = discriminator(P, a, b)
    P2 = dict(P)
    P2.update(u1)
    pf, u2 = discriminator(P2, a, out)
    return torch.relu(1. - pr).mean() + torch.relu(1. + pf).mean(), u2


def g_loss(P, a, b, masks, ngf=64, gan_weight=1.0, l1_weight=100.0):
    """train.py:504-512; the real pass only advances u (its loss term is not differentiated w.r.t. g_vars)"""
    out = generator(P, a, masks, ngf)
    with torch.no_grad():
        _, u1 = discriminator(P, a, b)
    P2 = dict(P)
    P2.update(u1)
    pf, u2 = discriminator(P2, a, out)
    gan, l1 = -pf.mean(), (b - out).abs().mean()
    return gan * gan_weight + l1 * l1_weight, dict(gan=gan, l1=l1, out=out, u=u2)
