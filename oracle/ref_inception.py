"""TEST INFRASTRUCTURE (never imported by the product): float64 torch-CPU restatement of the classifier the reference's
Inception-score harness runs (common/inception/inception_score.py:29-47: tf.image.resize_bilinear to 299 x 299, then tfgan's
frozen 2015 Inception graph, output `logits:0`).  The graph is a download (absent here) and TensorFlow cannot be imported, so
this restates the PUBLISHED architecture of that graph ("Rethinking the Inception Architecture", the inception-2015-12-05
release: scopes conv .. conv_4, mixed .. mixed_10, pool_3, softmax) layer by layer with torch.nn.functional, batch norm
applied explicitly (inference statistics, epsilon 0.001) -- parity unpinned, like the rest of the oracle: no reference-held
vector exists for it.  tests/test_inception_gpu.py compares the HIP network with it on random weights.
"""
import torch
import torch.nn.functional as F

BN_EPS = 0.001


def resize_bilinear_tf1(x, size):
    """tf.image.resize_bilinear (TF 1.x, align_corners=False): source coordinate = destination * in / out, no half-pixel shift.
    x [N,C,H,W] float64."""
    n, c, h, w = x.shape

    def axis(n_in, n_out):
        s = torch.arange(n_out, dtype=torch.float64) * (n_in / n_out)
        i0 = torch.floor(s).long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, (s - i0.double())
    y0, y1, fy = axis(h, size)
    x0, x1, fx = axis(w, size)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None]


class Net:
    def __init__(self, params):
        self.p = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in params.items()}

    def cbr(self, x, scope, stride=1, valid=False):
        """conv (no bias) -> batch norm (moving statistics) -> relu; filter [kh,kw,Cin,Cout] as TensorFlow stores it"""
        w = self.p[scope + '/conv2d_params']
        kh, kw = w.shape[0], w.shape[1]
        pad = (0, 0) if valid else ((kh - 1) // 2, (kw - 1) // 2)
        y = F.conv2d(x, w.permute(3, 2, 0, 1), None, stride, pad)
        g = self.p.get(scope + '/batchnorm/gamma')
        mean, var, beta = (self.p[f'{scope}/batchnorm/{s}'] for s in ('moving_mean', 'moving_variance', 'beta'))
        y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
        if g is not None:
            y = y * g[None, :, None, None]
        return torch.relu(y + beta[None, :, None, None])

    @staticmethod
    def avg3(x):
        return F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)          # tf.nn.avg_pool SAME: the padding is not counted

    def features(self, images, resize=299):
        """images [N,H,W,3] in [-1,1] -> pool_3 [N, 2048]"""
        x = torch.as_tensor(images, dtype=torch.float64).permute(0, 3, 1, 2)
        if resize and (x.shape[2] != resize or x.shape[3] != resize):
            x = resize_bilinear_tf1(x, resize)
        c = self.cbr
        x = c(x, 'conv', 2, True)
        x = c(x, 'conv_1', 1, True)
        x = c(x, 'conv_2')
        x = F.max_pool2d(x, 3, 2)
        x = c(x, 'conv_3', 1, True)
        x = c(x, 'conv_4', 1, True)
        x = F.max_pool2d(x, 3, 2)
        for m in ('mixed', 'mixed_1', 'mixed_2'):
            x = torch.cat([c(x, m + '/conv'),
                           c(c(x, m + '/tower/conv'), m + '/tower/conv_1'),
                           c(c(c(x, m + '/tower_1/conv'), m + '/tower_1/conv_1'), m + '/tower_1/conv_2'),
                           c(self.avg3(x), m + '/tower_2/conv')], 1)
        x = torch.cat([c(x, 'mixed_3/conv', 2, True),
                       c(c(c(x, 'mixed_3/tower/conv'), 'mixed_3/tower/conv_1'), 'mixed_3/tower/conv_2', 2, True),
                       F.max_pool2d(x, 3, 2)], 1)
        for m in ('mixed_4', 'mixed_5', 'mixed_6', 'mixed_7'):
            t1 = c(c(c(x, m + '/tower/conv'), m + '/tower/conv_1'), m + '/tower/conv_2')
            t2 = c(x, m + '/tower_1/conv')
            for j in (1, 2, 3, 4):
                t2 = c(t2, f'{m}/tower_1/conv_{j}')
            x = torch.cat([c(x, m + '/conv'), t1, t2, c(self.avg3(x), m + '/tower_2/conv')], 1)
        t2 = c(c(c(x, 'mixed_8/tower_1/conv'), 'mixed_8/tower_1/conv_1'), 'mixed_8/tower_1/conv_2')
        x = torch.cat([c(c(x, 'mixed_8/tower/conv'), 'mixed_8/tower/conv_1', 2, True),
                       c(t2, 'mixed_8/tower_1/conv_3', 2, True),
                       F.max_pool2d(x, 3, 2)], 1)
        for m, pool in (('mixed_9', self.avg3), ('mixed_10', lambda t: F.max_pool2d(t, 3, 1, 1))):
            a = c(x, m + '/tower/conv')
            b = c(c(x, m + '/tower_1/conv'), m + '/tower_1/conv_1')
            x = torch.cat([c(x, m + '/conv'),
                           c(a, m + '/tower/mixed/conv'), c(a, m + '/tower/mixed/conv_1'),
                           c(b, m + '/tower_1/mixed/conv'), c(b, m + '/tower_1/mixed/conv_1'),
                           c(pool(x), m + '/tower_2/conv')], 1)
        return x.mean(dim=(2, 3))

    def logits(self, images, resize=299):
        return self.features(images, resize) @ self.p['softmax/weights'] + self.p['softmax/biases']
