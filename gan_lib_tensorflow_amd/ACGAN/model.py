"""ACGAN ResNet for CIFAR-10 -- drop-in for ACGAN/model.py of the reference (BASELINE.json config 3).

`ACGAN().get_generator(z_var, labels, training, reuse)` and `.get_discriminator(x_var, labels, update_collection, reuse)`
keep the reference signatures (model.py:21,49), scopes (`g_net`, `d_net`) and variable names.

  * generator (model.py:31-47): the common block library (`ResidualBlock(resample='up', labels=...)`: conditional batch
    norm + relu, `G.OutputN` unconditional batch norm) -- the fused first-order kernels of the SNGAN path;
  * critic (model.py:59-88): no spectral norm; `Normalize` resolves to train-mode batch norm for the residual blocks
    (common/resnet_block.py:32-39 with spectral_normed=False), leaky-relu 0.2, two heads (GAN logit, 10-way classifier).
    Its loss carries the WGAN-GP term (train.py:99-107), so every operator is taken from `functional2` (twice
    differentiable); batch-norm moving statistics are updated by one kernel per call.
Tensors are bf16 NHWC; images are [N, 32, 32, 3] as in the reference (train.py:83).
"""
import numpy as np

from .. import functional2 as F2
from .. import kernels as K
from ..common import resnet_block as blocks
from ..common.ops import conv2d as _conv2d
from ..common.ops import linear as _linear
from ..store import get_default_store


def _conv(x, name, cin, cout, k, he_init=True):
    w, b = _conv2d.conv2d_variables(cin, cout, k, 1, name, he_init=he_init, biases=True)
    return F2.conv2d(x, w, b)


def _linear_vars(name, cin, cout):
    store = get_default_store()
    with store.variable_scope(name):
        std = np.sqrt(2. / (cin + cout))
        w = store.get_variable('W', [cin, cout], lambda rng: rng.uniform(-std * np.sqrt(3), std * np.sqrt(3), size=(cin, cout)).astype('float32'))
        b = store.get_variable('b', [cout], np.zeros(cout, 'float32'))
    return w, b


def _batch_norm(x, name, decay=0.9):
    """Normalize(name, x, spectral_normed=False) of common/resnet_block.py:32-39 -> normalization.batch_norm (:8-24)"""
    store = get_default_store()
    c = x.shape[-1]
    with store.variable_scope(name):
        with store.variable_scope('BatchNorm'):
            beta = store.get_variable('beta', [1, c], np.zeros((1, c), 'float32'))
            gamma = store.get_variable('gamma', [1, c], np.ones((1, c), 'float32'))
            mm = store.get_variable('moving_mean', [c], np.zeros(c, 'float32'), trainable=False)
            mv = store.get_variable('moving_variance', [c], np.ones(c, 'float32'), trainable=False)
            with store.variable_scope('moving_mean'):
                biased = store.get_variable('biased', [c], np.zeros(c, 'float32'), trainable=False)
                step = store.get_variable('local_step', [1], np.zeros(1, 'float32'), trainable=False)
    y, stats = F2.batch_norm_train(x, gamma, beta)
    K.bn_moving_update(stats.view(1, 2, c), mm, mv, biased, step, x.numel() // c, decay)
    return y


def _residual_block(x, dim, name, resample):
    """ResidualBlock(..., spectral_normed=False, activation_fn='lrelu') of common/resnet_block.py:100-156, resample 'down' | None"""
    xs, x = F2.fork(x)                       # the block input feeds the shortcut and the main path
    if resample == 'down':
        shortcut = F2.meanpool2x2(_conv(xs, name + '.Shortcut', dim, dim, 1, he_init=False))      # ConvMeanPool, filter_size 1
    elif resample is None:
        shortcut = xs                                                                           # identity skip-connection
    else:
        raise Exception('invalid resample value')
    h = F2.lrelu(_batch_norm(x, name + '.N1'))
    h = _conv(h, name + '.Conv1', dim, dim, 3)
    h = F2.lrelu(_batch_norm(h, name + '.N2'))
    h = _conv(h, name + '.Conv2', dim, dim, 3)
    if resample == 'down':
        h = F2.meanpool2x2(h)
    return F2.add(shortcut, h)


class ACGAN(object):
    def __init__(self):
        pass

    def get_generator(self, z_var, labels=None, training=True, reuse=False):
        """g-net (model.py:21-47): z [N, z_dim] bf16 -> images [N, 32, 32, 3] bf16 in tanh range"""
        store = get_default_store()
        with store.variable_scope('g_net', reuse=reuse):
            z = z_var.reshape(z_var.shape[0], -1)
            output = _linear.Linear(z, z.shape[-1], 4 * 4 * 1024, 'G.Input')
            output = output.reshape(-1, 4, 4, 1024)
            output = blocks.ResidualBlock(output, 1024, 256, 3, 'G.1', resample='up', labels=labels)
            output = blocks.ResidualBlock(output, 256, 256, 3, 'G.2', resample='up', labels=labels)
            output = blocks.ResidualBlock(output, 256, 256, 3, 'G.3', resample='up', labels=labels)
            output = blocks.Normalize('G.OutputN', output, relu=True)               # batch norm + nonlinearity (:42-43)
            output = _conv2d.Conv2D(output, 256, 3, 3, 1, 'G.Output', he_init=False, biases=True, out_tanh=True)
            return output

    def get_discriminator(self, x_var, labels=None, update_collection=None, reuse=False):
        """d-net (model.py:49-90): images [N, 32, 32, 3] -> (logits [N], class logits [N, 10])"""
        store = get_default_store()
        with store.variable_scope('d_net', reuse=reuse):
            # OptimizedResBlockDisc1(x, activation_fn='lrelu')   (resnet_block.py:159-184)
            x_short, x_var = F2.fork(x_var)     # (only when the input carries a gradient: the interpolates of the penalty term)
            shortcut = _conv(F2.meanpool2x2(x_short), 'D.DownBlock.1.Shortcut', 3, 128, 1, he_init=False)    # MeanPoolConv
            h = _conv(x_var, 'D.DownBlock.1.Conv1', 3, 128, 3)
            h = F2.lrelu(h)
            h = F2.meanpool2x2(_conv(h, 'D.DownBlock.1.Conv2', 128, 128, 3))                                 # ConvMeanPool
            output = F2.add(shortcut, h)
            output = _residual_block(output, 128, 'D.DownBlock.2', 'down')
            output = _residual_block(output, 128, 'D.NoneBlock.3', None)
            output = _residual_block(output, 128, 'D.NoneBlock.4', None)
            output = F2.lrelu(output)
            output = F2.mean_hw(output)                                                                      # reduce_mean(axis=[1, 2])
            out_w, out_c = F2.fork(output)      # the pooled features feed both heads
            w, b = _linear_vars('D.Output', 128, 1)
            output_wgan = F2.linear(out_w, w, b).reshape(-1)
            w, b = _linear_vars('D.ACGANOutput', 128, 10)
            output_acgan = F2.linear(out_c, w, b)
            return output_wgan, output_acgan
