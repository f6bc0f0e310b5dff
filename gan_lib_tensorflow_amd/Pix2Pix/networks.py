"""Pix2Pix U-Net generator and PatchGAN critic -- drop-in for the U-Net section of Pix2Pix/networks.py of the reference
(:355-536; BASELINE.json config 5).  Same function names, arguments, scopes (`encoder_%d`, `decoder_%d`, `layer_%d`) and
variable names; tensors are bf16 NHWC.

MI355X-first: the tf.pad of the critic (:482,503,523) and the nearest-neighbour upsampling of the decoder (`concat x4 +
depth_to_space` or `resize_images(NEAREST)`, :428-434 -- the same function) are never materialised: both are index arithmetic
in the gather of the convolution that consumes them, as is the decoder's relu; the skip concatenations are one launch each.
"""
from .. import functional as Fn
from ..common.ops import conv2d as _conv2d
from ..common.ops import normalization as _norm
from ..store import get_default_store


def nonlinearity(x, activation_fn='relu', leakiness=0.2):
    """networks.py:10-22"""
    if activation_fn == 'relu':
        return Fn.relu(x, 0.0)
    if activation_fn == 'lrelu':
        assert 0 < leakiness <= 1, "leakiness must be <= 1"
        return Fn.relu(x, leakiness)
    raise NotImplementedError('activation function [%s] is not recognized' % activation_fn)


def norm_layer(inputs, decay=0.9, epsilon=1e-5, is_training=True, norm_type="BN"):
    """networks.py:25-43"""
    if norm_type == "BN":
        return _norm.batch_norm(inputs, decay=decay, epsilon=epsilon, is_training=True)
    if norm_type == "IN":
        return _norm.instance_norm(inputs, epsilon=epsilon)
    raise NotImplementedError('Normalization [%s] is not implemented!' % norm_type)


def _check(conv_type, upsampe_method=None):
    if conv_type != 'conv2d':
        raise NotImplementedError('{0} is not supported!'.format(conv_type))
    if upsampe_method is not None and upsampe_method not in ('resize', 'depth_to_space'):
        raise NotImplementedError('upsampe_method [%s] is not recognized' % upsampe_method)


def unet_generator(generator_inputs, generator_outputs_channels, ngf, conv_type, channel_multiplier, padding,
                   upsampe_method='depth_to_space', rng_state=None):
    """networks.py:359-468.  rng_state: device RNG state for the decoder's dropout (kernels.new_rng_state)."""
    _check(conv_type, upsampe_method)
    store = get_default_store()
    layers = []
    with store.variable_scope("encoder_1"):
        cin = generator_inputs.shape[-1]
        output = _conv2d.Conv2D(generator_inputs, cin, ngf, 4, 2, 'Conv2D', conv_type=conv_type, padding=padding, he_init=True, biases=True)
        layers.append(output)
    # Every encoder output but the last feeds the next encoder AND a decoder's skip concat: the fan-out is explicit (Fn.fork: two
    # aliases forward, ONE add launch of the library backward) so that autograd's own accumulation never runs.  `skips[k]` is the
    # alias the decoder reads, `layers[k]` the one the next encoder reads.
    skips = {}
    for out_channels in (ngf * 2, ngf * 4, ngf * 8, ngf * 8, ngf * 8, ngf * 8, ngf * 8, ngf * 8):
        with store.variable_scope("encoder_%d" % (len(layers) + 1)):
            layers[-1], skips[len(layers) - 1] = Fn.fork(layers[-1])
            rectified = nonlinearity(layers[-1], 'lrelu', 0.2)
            convolved = _conv2d.Conv2D(rectified, rectified.shape[-1], out_channels, 4, 2, 'Conv2D', conv_type=conv_type,
                                       padding=padding, he_init=True, biases=True)
            layers.append(norm_layer(convolved, decay=0.9, epsilon=1e-5, is_training=True, norm_type="IN"))
    layer_specs = [(ngf * 8, 0.5), (ngf * 8, 0.5), (ngf * 8, 0.5), (ngf * 8, 0.0), (ngf * 8, 0.0), (ngf * 4, 0.0), (ngf * 2, 0.0), (ngf, 0.0)]
    num_encoder_layers = len(layers)
    for decoder_layer, (out_channels, dropout) in enumerate(layer_specs):
        skip_layer = num_encoder_layers - decoder_layer - 1
        with store.variable_scope("decoder_%d" % (skip_layer + 1)):
            # first decoder layer doesn't have skip connections since it is directly connected to the skip_layer
            inputs = layers[-1] if decoder_layer == 0 else Fn.concat_channels(layers[-1], skips[skip_layer])
            # relu, 2x nearest-neighbour upsampling and the 4x4 SAME convolution: one gather (:424-439)
            output = _conv2d.Conv2D(inputs, inputs.shape[-1], out_channels, 4, 1, 'Conv2D', conv_type=conv_type, padding=padding,
                                    he_init=True, biases=True, upsample=True, in_relu=True)
            output = norm_layer(output, decay=0.9, epsilon=1e-5, is_training=True, norm_type="IN")
            if dropout > 0.0:
                if rng_state is None:
                    raise ValueError('the decoder drops out: unet_generator needs an rng_state')
                output = Fn.dropout(output, 1 - dropout, rng_state)
            layers.append(output)
    with store.variable_scope("decoder_1"):
        inputs = Fn.concat_channels(layers[-1], skips[0])
        output = _conv2d.Conv2D(inputs, inputs.shape[-1], generator_outputs_channels, 4, 1, 'Conv2D', conv_type=conv_type, padding=padding,
                                he_init=True, biases=True, upsample=True, in_relu=True, out_tanh=True)
        layers.append(output)
    return layers[-1]


def unet_discriminator(discrim_inputs, discrim_targets, ndf, spectral_normed, update_collection, conv_type, channel_multiplier, padding):
    """networks.py:471-536: 70x70 PatchGAN on concat(inputs, targets): tf.pad 1 + 4x4 VALID convs, stride 2 x4 then 1 x2"""
    _check(conv_type)
    if padding != 'VALID':
        raise NotImplementedError('the critic pads explicitly and convolves VALID (train.py:463,474)')
    store = get_default_store()
    n_layers = 4
    inputs = Fn.concat_channels(discrim_inputs, discrim_targets)
    kw = dict(conv_type=conv_type, padding=padding, spectral_normed=spectral_normed, update_collection=update_collection,
              he_init=True, biases=True, pad_input=1)
    with store.variable_scope("layer_1"):
        rectified = nonlinearity(_conv2d.Conv2D(inputs, inputs.shape[-1], ndf, 4, 2, 'Conv2D', **kw), 'lrelu', 0.2)
    layers = [rectified]
    for i in range(n_layers):
        with store.variable_scope("layer_%d" % (len(layers) + 1)):
            out_channels_ = ndf * min(2 ** (i + 1), 8)
            stride = 1 if i == n_layers - 1 else 2          # last layer here has stride 1
            convolved = _conv2d.Conv2D(layers[-1], layers[-1].shape[-1], out_channels_, 4, stride, 'Conv2D', **kw)
            rectified = nonlinearity(convolved, 'lrelu', 0.2)
            layers.append(rectified)
    with store.variable_scope("layer_%d" % (len(layers) + 1)):
        output = _conv2d.Conv2D(rectified, rectified.shape[-1], 1, 4, 1, 'Conv2D', **kw)
        layers.append(output)
    return layers[-1]
