"""Transposed convolution -- drop-in for common/ops/deconv2d.py of the reference.

tf.nn.conv2d_transpose, stride 2, SAME, filter [k,k,Cout,Cin], output 2H x 2W (deconv2d.py:99-114).
The op has no caller in the reference; it is provided at op level on the MFMA engines: fprop by output phase (four
2x2-tap convolutions over the low-resolution input, 4 MACs per output) for filter sizes 3 and 4 with in_channels % 64 == 0,
the zero-insertion gather otherwise (k = 5: three taps per axis in the odd phases); stride-2 gather for dgrad / wgrad."""
import numpy as np
import torch
from torch.autograd import Function

from ... import kernels as K
from ...functional import _target, _c
from ...store import get_default_store

_default_weightnorm = False
_weights_stdev = None
PHASE_FORM = True    # fprop by output phase where it applies (no MACs on inserted zeros)


class _Deconv2d(Function):
    @staticmethod
    def forward(ctx, x, F, bias):
        k, _, cout, cin = F.shape
        b = bias.detach() if bias is not None else None
        if PHASE_FORM and k in (3, 4) and cin % 64 == 0:
            wfz, _ = K.prep_weights(F.detach(), True, False)  # the dgrad operand: F viewed as HWIO (I=Cout, O=Cin)
            y = K.upconv3x3_fprop(x, K.deconv2d_prep_phases(F.detach()), b, cout)
        else:
            wfz, wz = K.prep_weights(F.detach(), True, True)
            y = K.deconv2d_fprop(x, wz, b, cout, k)
        ctx.save_for_backward(x, F, wfz)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, F, wfz = ctx.saved_tensors
        k, _, cout, cin = F.shape
        g = _c(dy)
        dx = dF = db = None
        if ctx.needs_input_grad[0]:
            dx = K.deconv2d_dgrad(g, wfz, cin, k)
        if ctx.needs_input_grad[1]:
            tgt, acc = _target(F)
            K.deconv2d_wgrad(x, g, tgt, k)
            dF = None if acc else tgt
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            tgt, acc = _target(ctx.bias)
            K.colsum(g, tgt, 1.0)
            db = None if acc else tgt
        return dx, dF, db


def Deconv2D(inputs, in_channels, output_channels, filter_size, stride=2, padding='SAME', he_init=True,
             weight_norm=None, gain=1., mask_type=None, biases=True, name='Deconv2D'):
    """inputs [batch, height, width, in_channels] -> [batch, 2*height, 2*width, output_channels]"""
    store = get_default_store()
    with store.variable_scope(name):
        if mask_type is not None:
            raise Exception('Unsupported configuration in Deconv2D!')
        if stride != 2 or padding != 'SAME' or (weight_norm if weight_norm is not None else _default_weightnorm):
            raise NotImplementedError('Deconv2D is built for stride 2, SAME, no weight norm (deconv2d.py:29-30)')

        def init(rng):
            fan_in = in_channels * filter_size ** 2 / (stride ** 2)
            fan_out = output_channels * filter_size ** 2
            if _weights_stdev is not None:
                stdev = _weights_stdev
            elif he_init:
                stdev = np.sqrt(4. / (fan_in + fan_out))
            else:
                stdev = np.sqrt(2. / (fan_in + fan_out))
            return gain * rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3),
                                      size=(filter_size, filter_size, output_channels, in_channels)).astype('float32')

        filters = store.get_variable('Filters', [filter_size, filter_size, output_channels, in_channels], init)
        _biases = store.get_variable('Biases', [output_channels], np.zeros(output_channels, 'float32')) if biases else None
        return _Deconv2d.apply(inputs, filters, _biases)
