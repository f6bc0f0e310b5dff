"""Convolution for NHWC data -- drop-in for common/ops/conv2d.py of the reference (conv_type='conv2d').

Same signature, variable names (`<name>/Filters`, `<name>/Biases`, `<name>/filters/spectral_norm/u`),
initialisation (conv2d.py:83-140) and error behaviour; the TensorFlow ops underneath
(conv2d.py:180-187, 212-216) are replaced by the MFMA implicit-GEMM kernels of libgank.so.
Extra keyword arguments (`residual`, `upsample`, `in_relu`, `pool_out`, `out_tanh`) expose the
fusions the block library uses; they default to the reference behaviour.
"""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store
from .sn import spectral_normed_weight

_default_weightnorm = False


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


_weights_stdev = None


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def conv2d_variables(input_dim, output_dim, filter_size=3, stride=1, name='Conv2D', conv_type='conv2d', padding='SAME',
                     spectral_normed=False, update_collection=None, inputs_norm=False, he_init=True,
                     mask_type=None, weightnorm=None, biases=True, gain=1.):
    """The variable half of Conv2D (conv2d.py:59-176): creates / fetches `<name>/Filters` (spectrally normalised when
    asked) and `<name>/Biases` under the current scope and returns (filters, biases | None).  Shared by Conv2D and by
    fused multi-convolution kernels, so both paths own identical variables."""
    store = get_default_store()
    with store.variable_scope(name):
        if conv_type != 'conv2d':
            # depthwise / separable branches (conv2d.py:188-208) are not on the SNGAN hot path
            raise NotImplementedError('{0} is not supported!'.format(conv_type))
        if mask_type is not None or inputs_norm or (weightnorm if weightnorm is not None else _default_weightnorm):
            raise NotImplementedError('mask_type / inputs_norm / weightnorm are outside the SNGAN hot path')
        if stride not in (1, 2) or padding not in ('SAME', 'VALID'):
            raise NotImplementedError('stride 1 | 2, SAME | VALID (conv2d.py:180-187)')

        def init(rng):
            fan_in = input_dim * filter_size ** 2
            fan_out = output_dim * filter_size ** 2 / (stride ** 2)
            if _weights_stdev is not None:
                stdev = _weights_stdev
            elif he_init:
                stdev = np.sqrt(4. / (fan_in + fan_out))
            else:  # Normalized init (Glorot & Bengio)
                stdev = np.sqrt(2. / (fan_in + fan_out))
            vals = rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3),
                               size=(filter_size, filter_size, input_dim, output_dim)).astype('float32')
            return vals * gain

        filters = store.get_variable('Filters', [filter_size, filter_size, input_dim, output_dim], init)

        if spectral_normed:
            with store.variable_scope('filters'):
                filters = spectral_normed_weight(filters, update_collection=update_collection)

        _biases = None
        if biases:
            _biases = store.get_variable('Biases', [output_dim], np.zeros(output_dim, 'float32'))
        return filters, _biases


def Conv2D(inputs, input_dim, output_dim, filter_size=3, stride=1, name='Conv2D',
           conv_type='conv2d', channel_multiplier=0, padding='SAME',
           spectral_normed=False, update_collection=None, inputs_norm=False, he_init=True,
           mask_type=None, weightnorm=None, biases=True, gain=1.,
           residual=None, upsample=False, in_relu=False, pool_out=False, out_tanh=False, stats_groups=0, pad_input=0):
    """inputs: bf16 tensor [batch, height, width, in_channels] on the GPU.
    Returns [batch, out_height, out_width, output_dim].  stats_groups: the result feeds a batch norm over that many
    towers (functional.conv2d).  Odd filters at stride 1 with SAME padding run on the fused SNGAN-path kernels; even
    filters, stride 2 and VALID padding on the general gather (functional.conv2d_general).  pad_input: zeros the caller
    would have added around `inputs` with tf.pad before a VALID convolution (Pix2Pix/networks.py:482,503,523) -- they are
    never materialised."""
    filters, _biases = conv2d_variables(input_dim, output_dim, filter_size, stride, name, conv_type, padding,
                                        spectral_normed, update_collection, inputs_norm, he_init, mask_type, weightnorm,
                                        biases, gain)
    if stride != 1 or padding != 'SAME' or filter_size % 2 == 0 or pad_input:
        if residual is not None or pool_out or stats_groups:
            raise NotImplementedError('residual / pool_out / stats_groups belong to the stride-1 SAME kernels')
        h, w = inputs.shape[1] * (2 if upsample else 1), inputs.shape[2] * (2 if upsample else 1)
        if padding == 'SAME':                       # tf.nn.conv2d: out = ceil(in / stride), surplus pad at the bottom / right
            if pad_input:
                raise NotImplementedError('pad_input goes with VALID padding')
            oh, ow = -(-h // stride), -(-w // stride)
            pad = max((oh - 1) * stride + filter_size - h, 0) // 2
            assert pad == max((ow - 1) * stride + filter_size - w, 0) // 2, 'SAME pad differs between the axes'
        else:
            oh, ow = (h + 2 * pad_input - filter_size) // stride + 1, (w + 2 * pad_input - filter_size) // stride + 1
            pad = pad_input
        return Fn.conv2d_general(inputs, filters, _biases, stride=stride, pad=pad, out_hw=(oh, ow), upsample=upsample,
                                 in_relu=in_relu, out_tanh=out_tanh)
    return Fn.conv2d(inputs, filters, _biases, residual=residual, upsample=upsample, in_relu=in_relu,
                     pool_out=pool_out, out_tanh=out_tanh, stats_groups=stats_groups)
