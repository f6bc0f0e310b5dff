"""The classifier of the Inception-score harness on the HIP kernels.

common/inception/inception_score.py:29-47 of the reference resizes the samples to 299 x 299 (tf.image.resize_bilinear) and runs
tfgan's `run_inception(..., output_tensor='logits:0')`, i.e. the frozen 2015 Inception graph ("inception-2015-12-05",
Inception-v3: node scopes conv .. conv_4, mixed .. mixed_10, pool_3, softmax; 1008 logits, the reference keeps the first 1000,
:53).  That graph is a download and cannot be fetched here, so this module is the NETWORK with a weights-file loader:
`InceptionV3(params)` takes a dict of arrays under the graph's own node names and runs the forward pass on
gank_conv2d_general_fprop / gank_pool2d / gank_relu_to_channels / gank_resize_bilinear / gank_linear_fwd.  With the real
weights supplied (`InceptionV3.from_npz`: an .npz exported from the frozen graph's constants) the score of
`get_inception_score(images, classifier=net.logits)` is the reference's; without them the network is parity-tested with random
weights against the float64 restatement oracle/ref_inception.py (tests/test_inception_gpu.py).

Weights file (every array float32, TensorFlow layouts):
    <scope>/conv2d_params                      [kh, kw, Cin, Cout]   (e.g. mixed_4/tower/conv_1/conv2d_params is 1 x 7)
    <scope>/batchnorm/{beta,moving_mean,moving_variance}  [Cout]    (and optionally gamma; epsilon 0.001)
    softmax/weights [2048, 1008], softmax/biases [1008]
`param_shapes()` lists all of them.  Batch norm is folded into the filter and a bias at load time (inference statistics);
rectangular filters (1x7, 7x1, 1x3, 3x1) are embedded in the centre of a square one (the general conv kernel takes square
filters up to 7 x 7: the zero taps cost FLOPs, not accuracy).
"""
import numpy as np
import torch

from ... import kernels as K

BN_EPS = 0.001


def _block_35(name, pool_proj):
    return [(f'{name}/conv', 1, 1, 64), (f'{name}/tower/conv', 1, 1, 48), (f'{name}/tower/conv_1', 5, 5, 64),
            (f'{name}/tower_1/conv', 1, 1, 64), (f'{name}/tower_1/conv_1', 3, 3, 96), (f'{name}/tower_1/conv_2', 3, 3, 96),
            (f'{name}/tower_2/conv', 1, 1, pool_proj)]


def _block_17(name, c):
    return [(f'{name}/conv', 1, 1, 192), (f'{name}/tower/conv', 1, 1, c), (f'{name}/tower/conv_1', 1, 7, c), (f'{name}/tower/conv_2', 7, 1, 192),
            (f'{name}/tower_1/conv', 1, 1, c), (f'{name}/tower_1/conv_1', 7, 1, c), (f'{name}/tower_1/conv_2', 1, 7, c),
            (f'{name}/tower_1/conv_3', 7, 1, c), (f'{name}/tower_1/conv_4', 1, 7, 192), (f'{name}/tower_2/conv', 1, 1, 192)]


def _block_8(name):
    return [(f'{name}/conv', 1, 1, 320), (f'{name}/tower/conv', 1, 1, 384), (f'{name}/tower/mixed/conv', 1, 3, 384), (f'{name}/tower/mixed/conv_1', 3, 1, 384),
            (f'{name}/tower_1/conv', 1, 1, 448), (f'{name}/tower_1/conv_1', 3, 3, 384), (f'{name}/tower_1/mixed/conv', 1, 3, 384),
            (f'{name}/tower_1/mixed/conv_1', 3, 1, 384), (f'{name}/tower_2/conv', 1, 1, 192)]


def conv_layers():
    """[(scope, kh, kw, Cin, Cout)] of every convolution, in graph order"""
    out = []
    cin = {}

    def add(specs, inputs):
        for (nm, kh, kw, co), ci in zip(specs, inputs):
            out.append((nm, kh, kw, ci, co))
            cin[nm] = co
    add([('conv', 3, 3, 32), ('conv_1', 3, 3, 32), ('conv_2', 3, 3, 64), ('conv_3', 1, 1, 80), ('conv_4', 3, 3, 192)], [3, 32, 32, 64, 80])
    c = 192
    for nm, pp in (('mixed', 32), ('mixed_1', 64), ('mixed_2', 64)):
        add(_block_35(nm, pp), [c, c, 48, c, 64, 96, c])
        c = 64 + 64 + 96 + pp
    add([('mixed_3/conv', 3, 3, 384), ('mixed_3/tower/conv', 1, 1, 64), ('mixed_3/tower/conv_1', 3, 3, 96), ('mixed_3/tower/conv_2', 3, 3, 96)], [c, c, 64, 96])
    c = 384 + 96 + c          # 768
    for nm, w in (('mixed_4', 128), ('mixed_5', 160), ('mixed_6', 160), ('mixed_7', 192)):
        add(_block_17(nm, w), [c, c, w, w, c, w, w, w, w, c])
    add([('mixed_8/tower/conv', 1, 1, 192), ('mixed_8/tower/conv_1', 3, 3, 320), ('mixed_8/tower_1/conv', 1, 1, 192), ('mixed_8/tower_1/conv_1', 1, 7, 192),
         ('mixed_8/tower_1/conv_2', 7, 1, 192), ('mixed_8/tower_1/conv_3', 3, 3, 192)], [c, 192, c, 192, 192, 192])
    c = 320 + 192 + c         # 1280
    for nm in ('mixed_9', 'mixed_10'):
        add(_block_8(nm), [c, c, 384, 384, c, 448, 384, 384, c])
        c = 320 + 768 + 768 + 192      # 2048
    return out


def param_shapes():
    """{name: shape} of the weights file"""
    shapes = {}
    for nm, kh, kw, ci, co in conv_layers():
        shapes[nm + '/conv2d_params'] = (kh, kw, ci, co)
        for s in ('beta', 'moving_mean', 'moving_variance'):
            shapes[f'{nm}/batchnorm/{s}'] = (co,)
    shapes['softmax/weights'] = (2048, 1008)
    shapes['softmax/biases'] = (1008,)
    return shapes


class InceptionV3:
    def __init__(self, params, device='cuda'):
        self.device = torch.device(device)
        want = param_shapes()
        missing = [k for k in want if k not in params]
        if missing:
            raise KeyError(f"Inception weights: {len(missing)} arrays missing, e.g. {missing[:3]}")
        self.convs = {}
        todo = []
        for nm, kh, kw, ci, co in conv_layers():
            w = np.asarray(params[nm + '/conv2d_params'], np.float64)
            if w.shape != (kh, kw, ci, co):
                raise ValueError(f"{nm}/conv2d_params: shape {w.shape}, expected {(kh, kw, ci, co)}")
            gamma = np.asarray(params.get(f'{nm}/batchnorm/gamma', np.ones(co)), np.float64)
            scale = gamma / np.sqrt(np.asarray(params[f'{nm}/batchnorm/moving_variance'], np.float64) + BN_EPS)
            bias = np.asarray(params[f'{nm}/batchnorm/beta'], np.float64) - np.asarray(params[f'{nm}/batchnorm/moving_mean'], np.float64) * scale
            k = max(kh, kw)
            sq = np.zeros((k, k, ci, co))
            sq[(k - kh) // 2:(k - kh) // 2 + kh, (k - kw) // 2:(k - kw) // 2 + kw] = w * scale      # batch norm folded (inference statistics)
            wt = torch.tensor(sq.astype(np.float32), device=self.device)
            self.convs[nm] = [k, co, wt, torch.tensor(bias.astype(np.float32), device=self.device), None]
            todo.append(nm)
        for i in range(0, len(todo), 16):         # MFMA operand copies, 16 filters per launch
            part = todo[i:i + 16]
            outs = K.prep_weights_batched([self.convs[nm][2] for nm in part], want_d=False, kinds=[0] * len(part))
            for nm, (wf, _) in zip(part, outs):
                self.convs[nm][4] = wf
        self.fc_w = torch.tensor(np.asarray(params['softmax/weights'], np.float32), device=self.device)
        self.fc_b = torch.tensor(np.asarray(params['softmax/biases'], np.float32), device=self.device)

    @classmethod
    def from_npz(cls, path, device='cuda'):
        with np.load(path) as f:
            return cls({k: f[k] for k in f.files}, device)

    # ---- operators ------------------------------------------------------------------------------------------------------
    def _conv(self, x, nm, stride=1, valid=False, out=None, c_off=0):
        """conv + folded batch norm + ReLU; the ReLU writes into channels [c_off, ..) of `out` when given (branch concat)"""
        k, co, _, bias, wf = self.convs[nm]
        n, h, w, _ = x.shape
        if valid:
            pad, oh, ow = 0, (h - k) // stride + 1, (w - k) // stride + 1
        else:
            assert stride == 1
            pad, oh, ow = (k - 1) // 2, h, w
        y = K.conv2d_general_fprop(x, wf, bias, (oh, ow), co, k, stride, pad, 0)
        return K.relu_to_channels(y, out, c_off)

    def _new(self, x, hw, c):
        return torch.empty((x.shape[0], hw[0], hw[1], c), dtype=K.BF16, device=x.device)

    def _b35(self, x, nm):
        n, h, w, c = x.shape
        pp = self.convs[nm + '/tower_2/conv'][1]
        out = self._new(x, (h, w), 64 + 64 + 96 + pp)
        self._conv(x, nm + '/conv', out=out, c_off=0)
        self._conv(self._conv(x, nm + '/tower/conv'), nm + '/tower/conv_1', out=out, c_off=64)
        t = self._conv(self._conv(x, nm + '/tower_1/conv'), nm + '/tower_1/conv_1')
        self._conv(t, nm + '/tower_1/conv_2', out=out, c_off=128)
        self._conv(K.pool2d(x, 3, 1, 1, (h, w), 'avg'), nm + '/tower_2/conv', out=out, c_off=224)
        return out

    def _b17(self, x, nm):
        n, h, w, c = x.shape
        out = self._new(x, (h, w), 768)
        self._conv(x, nm + '/conv', out=out, c_off=0)
        t = self._conv(self._conv(x, nm + '/tower/conv'), nm + '/tower/conv_1')
        self._conv(t, nm + '/tower/conv_2', out=out, c_off=192)
        t = self._conv(x, nm + '/tower_1/conv')
        for j in (1, 2, 3):
            t = self._conv(t, f'{nm}/tower_1/conv_{j}')
        self._conv(t, nm + '/tower_1/conv_4', out=out, c_off=384)
        self._conv(K.pool2d(x, 3, 1, 1, (h, w), 'avg'), nm + '/tower_2/conv', out=out, c_off=576)
        return out

    def _b8(self, x, nm, pool):
        n, h, w, c = x.shape
        out = self._new(x, (h, w), 2048)
        self._conv(x, nm + '/conv', out=out, c_off=0)
        t = self._conv(x, nm + '/tower/conv')
        self._conv(t, nm + '/tower/mixed/conv', out=out, c_off=320)
        self._conv(t, nm + '/tower/mixed/conv_1', out=out, c_off=704)
        t = self._conv(self._conv(x, nm + '/tower_1/conv'), nm + '/tower_1/conv_1')
        self._conv(t, nm + '/tower_1/mixed/conv', out=out, c_off=1088)
        self._conv(t, nm + '/tower_1/mixed/conv_1', out=out, c_off=1472)
        self._conv(K.pool2d(x, 3, 1, 1, (h, w), pool), nm + '/tower_2/conv', out=out, c_off=1856)
        return out

    @torch.no_grad()
    def features(self, images, resize=299):
        """images [N,H,W,3] in [-1, 1] (any float dtype, or the 16-bit activation dtype) -> pool_3 features bf16 [N, 2048]"""
        x = torch.as_tensor(images)
        if x.dtype != K.BF16:
            x = x.to(torch.float32).to(K.BF16)
        x = x.to(self.device).contiguous()
        assert x.dim() == 4 and x.shape[3] == 3, tuple(x.shape)
        if resize and (x.shape[1] != resize or x.shape[2] != resize):
            x = K.resize_bilinear(x, (resize, resize))                       # tf.image.resize_bilinear (inception_score.py:32)
        x = self._conv(x, 'conv', stride=2, valid=True)
        x = self._conv(x, 'conv_1', valid=True)
        x = self._conv(x, 'conv_2')
        h = (x.shape[1] - 3) // 2 + 1
        x = K.pool2d(x, 3, 2, 0, (h, h), 'max')
        x = self._conv(x, 'conv_3', valid=True)
        x = self._conv(x, 'conv_4', valid=True)
        h = (x.shape[1] - 3) // 2 + 1
        x = K.pool2d(x, 3, 2, 0, (h, h), 'max')
        for nm in ('mixed', 'mixed_1', 'mixed_2'):
            x = self._b35(x, nm)
        # mixed_3: grid reduction 35 -> 17
        h = (x.shape[1] - 3) // 2 + 1
        c = x.shape[3]
        out = self._new(x, (h, h), 384 + 96 + c)
        self._conv(x, 'mixed_3/conv', stride=2, valid=True, out=out, c_off=0)
        t = self._conv(self._conv(x, 'mixed_3/tower/conv'), 'mixed_3/tower/conv_1')
        self._conv(t, 'mixed_3/tower/conv_2', stride=2, valid=True, out=out, c_off=384)
        K.pool2d(x, 3, 2, 0, (h, h), 'max', out=out, c_off=480)
        x = out
        for nm in ('mixed_4', 'mixed_5', 'mixed_6', 'mixed_7'):
            x = self._b17(x, nm)
        # mixed_8: grid reduction 17 -> 8
        h = (x.shape[1] - 3) // 2 + 1
        c = x.shape[3]
        out = self._new(x, (h, h), 320 + 192 + c)
        self._conv(self._conv(x, 'mixed_8/tower/conv'), 'mixed_8/tower/conv_1', stride=2, valid=True, out=out, c_off=0)
        t = self._conv(x, 'mixed_8/tower_1/conv')
        t = self._conv(self._conv(t, 'mixed_8/tower_1/conv_1'), 'mixed_8/tower_1/conv_2')
        self._conv(t, 'mixed_8/tower_1/conv_3', stride=2, valid=True, out=out, c_off=320)
        K.pool2d(x, 3, 2, 0, (h, h), 'max', out=out, c_off=512)
        x = out
        x = self._b8(x, 'mixed_9', 'avg')
        x = self._b8(x, 'mixed_10', 'max')          # the 2015 graph pools this branch with a MAX pool
        g = x.shape[1]
        return K.pool2d(x, g, 1, 0, (1, 1), 'avg').reshape(x.shape[0], 2048)        # pool_3: 8 x 8 average at 299 x 299

    @torch.no_grad()
    def logits(self, images, resize=299):
        """-> float32 numpy [N, 1008] (`logits:0`): the callable get_inception_score(classifier=...) expects"""
        f = self.features(images, resize)
        # fp32 logits: the score exponentiates them (16-bit logits at |l| ~ 10 would be +-0.04, a few percent per probability)
        return K.linear_fwd(f, self.fc_w, self.fc_b, out_f32=True).cpu().numpy()
