"""Block library -- drop-in for common/resnet_block.py:24-184 and the private copy inside
SNGAN/gan_cifar_resnet.py:80-234 of the reference.

Same functions, signatures and variable names.  What differs is underneath: every resampling /
activation / shortcut-add around a convolution is folded into the MFMA conv kernel's gather or
epilogue instead of being a TensorFlow op that materialises a tensor:
  UpsampleConv  = conv with the nearest-neighbour 2x applied while staging the input patch
                  (tf.concat x4 + tf.depth_to_space, gan_cifar_resnet.py:143-145, never materialised);
  ConvMeanPool  = conv + 2x2 mean (the gradient of the pool is folded into dgrad/wgrad gathers);
  pre-activation relu = relu on the conv operand (D) or fused into the CBN apply kernel (G);
  shortcut + output   = residual add in the conv epilogue.
"""
import functools

import torch

from .. import functional as Fn
from .. import kernels as K
from ..store import get_default_store
from .ops import conv2d as _conv2d
from .ops import normalization as _normalization

# reference globals (common/resnet_block.py:20-21; SNGAN/gan_cifar_resnet.py:43-44,51-52 overrides them)
NORMALIZATION_G = True
NORMALIZATION_D = False
CONDITIONAL = True
ACGAN = False
DIM_D = 128

# A 1x1 convolution commutes with 2x2 mean pooling and with nearest-neighbour upsampling (both act per pixel /
# per channel): the shortcut convolutions run at the LOW resolution, 4x fewer MACs (exact in real
# arithmetic; bf16 rounding of the intermediate moves from before to after the resampling).
COMMUTE_1X1 = True
# the commuted 'up' shortcut stays at half resolution and is added, upsampled on the fly, in conv_2's epilogue
FUSE_SHORTCUT_UPSAMPLE = True
FUSE_FORK_POOL = True   # down blocks: fan-out and shortcut pool as one op (one unpool-add launch backward)
FUSE_LABEL_FORK = True   # critic: label concat + the next block's fan-out (fork_pool) as one op each way
FUSE_POOL_GATHER = True   # first critic block: the shortcut's 2x2 mean inside its 1x1 conv's gather
FUSE_IDENTITY_SHORTCUT_GRAD = True   # identity shortcut: its gradient is added by conv_1's input-gradient kernel


def nonlinearity(x, activation_fn='relu', leakiness=0.2):
    """gan_cifar_resnet.py:80-85"""
    if activation_fn == 'relu':
        return Fn.relu(x, 0.0)
    if activation_fn == 'lrelu':
        assert 0 < leakiness <= 1, "leakiness must be <= 1"
        return Fn.relu(x, leakiness)


def _normalize_kind(name, labels):
    """Dispatch of Normalize (gan_cifar_resnet.py:88-109): 'cbn' | 'bn' | 'ln' | None."""
    if not CONDITIONAL:
        labels = None
    if CONDITIONAL and ACGAN and ('D.' in name):
        labels = None
    if ('D.' in name) and NORMALIZATION_D:
        return 'ln'
    if ('G.' in name) and NORMALIZATION_G:
        return 'cbn' if labels is not None else 'bn'
    return None


def Normalize(name, inputs, labels=None, groups=1, relu=False):
    """Chooses between batchnorm, layernorm, their conditional variants, or nothing, depending on
    `name` and the global flags (gan_cifar_resnet.py:88-109).  `relu=True` fuses the nonlinearity
    that always follows (only honoured when a normalisation actually runs)."""
    store = get_default_store()
    with store.variable_scope(name):
        kind = _normalize_kind(name, labels)
        if kind == 'ln':
            return _normalization.layer_norm(name, [1, 2, 3], inputs)
        if kind == 'cbn':
            return _normalization.cond_batchnorm(name, [0, 1, 2], inputs, labels=labels, n_labels=10,
                                                 groups=groups, relu=relu)
        if kind == 'bn':
            return _normalization.batch_norm(inputs, fused=True, groups=groups, relu=relu)
        return inputs


def ConvMeanPool(inputs, output_dim, filter_size=3, stride=1, name=None,
                 spectral_normed=False, update_collection=None, inputs_norm=False,
                 he_init=True, biases=True, **fused):
    """conv, then tf.add_n of the four stride-2 slices / 4 (gan_cifar_resnet.py:112-122)"""
    if filter_size == 1 and COMMUTE_1X1 and not fused:      # mean_pool(conv1x1(x) + b) == conv1x1(mean_pool(x)) + b
        return MeanPoolConv(inputs, output_dim, filter_size, stride, name, spectral_normed=spectral_normed,
                            update_collection=update_collection, he_init=he_init, biases=biases, **fused)
    return _conv2d.Conv2D(inputs, inputs.shape[-1], output_dim, filter_size, stride, name,
                          spectral_normed=spectral_normed, update_collection=update_collection,
                          he_init=he_init, biases=biases, pool_out=True, **fused)


def MeanPoolConv(inputs, output_dim, filter_size=3, stride=1, name=None,
                 spectral_normed=False, update_collection=None, inputs_norm=False,
                 he_init=True, biases=True, **fused):
    """2x2 mean, then conv (gan_cifar_resnet.py:125-137)"""
    output = Fn.meanpool2x2(inputs)
    return _conv2d.Conv2D(output, output.shape[-1], output_dim, filter_size, stride, name,
                          spectral_normed=spectral_normed, update_collection=update_collection,
                          he_init=he_init, biases=biases, **fused)


def UpsampleConv(inputs, output_dim, filter_size=3, stride=1, name=None,
                 spectral_normed=False, update_collection=None, inputs_norm=False,
                 he_init=True, biases=True, **fused):
    """nearest-neighbour 2x (concat x4 + depth_to_space), then conv (gan_cifar_resnet.py:140-153)"""
    if filter_size == 1 and COMMUTE_1X1 and not fused:      # conv1x1(upsample(x)) + b == upsample(conv1x1(x) + b)
        low = _conv2d.Conv2D(inputs, inputs.shape[-1], output_dim, filter_size, stride, name,
                             spectral_normed=spectral_normed, update_collection=update_collection,
                             he_init=he_init, biases=biases)
        if FUSE_SHORTCUT_UPSAMPLE:
            low._up2x = True      # consumed as `residual=`: the next conv's epilogue adds it upsampled (no materialisation)
            return low
        return Fn.upsample_nn2x(low)
    return _conv2d.Conv2D(inputs, inputs.shape[-1], output_dim, filter_size, stride, name,
                          spectral_normed=spectral_normed, update_collection=update_collection,
                          he_init=he_init, biases=biases, upsample=True, **fused)


def ResidualBlock(inputs, input_dim, output_dim, filter_size, name,
                  spectral_normed=False, update_collection=None, inputs_norm=False,
                  resample=None, labels=None, biases=True, groups=1, out_stats=0, prefork=None):
    """resample: None, 'down', or 'up'  (gan_cifar_resnet.py:156-209).  out_stats: the block's output feeds a batch norm
    over that many towers (its statistics then come out of conv_2's epilogue where the kernel can produce them).
    prefork = (inputs, mean_pool2x2(inputs)): the producer of a 'down' block's input already made the block's fan-out
    (functional.concat_label_fork_pool); `inputs` is then ignored."""
    if resample == 'down':
        conv_1 = functools.partial(_conv2d.Conv2D, input_dim=input_dim, output_dim=input_dim)
        conv_2 = functools.partial(ConvMeanPool, output_dim=output_dim)
        conv_shortcut = ConvMeanPool
    elif resample == 'up':
        conv_1 = functools.partial(UpsampleConv, output_dim=output_dim)
        conv_shortcut = UpsampleConv
        conv_2 = functools.partial(_conv2d.Conv2D, input_dim=output_dim, output_dim=output_dim)
    elif resample is None:
        conv_shortcut = functools.partial(_conv2d.Conv2D, input_dim=input_dim)
        conv_1 = functools.partial(_conv2d.Conv2D, input_dim=input_dim, output_dim=output_dim)
        conv_2 = functools.partial(_conv2d.Conv2D, input_dim=output_dim, output_dim=output_dim)
    else:
        raise Exception('invalid resample value')

    pooled_short = resample == 'down' and COMMUTE_1X1 and FUSE_FORK_POOL
    if prefork is not None:
        assert pooled_short, "prefork is the (x, mean_pool2x2(x)) pair of a down-sampling block"
        x_main, x_short = prefork
    elif pooled_short:
        x_main, x_short = Fn.fork_pool(inputs)     # the 1x1 shortcut conv commutes with the pool: it runs on the pooled alias
    else:
        x_short, x_main = Fn.fork(inputs)
    if output_dim == input_dim and resample is None:
        shortcut = x_short  # Identity skip-connection
        if FUSE_IDENTITY_SHORTCUT_GRAD and _normalize_kind(name + '.N1', labels) is None and x_short is not x_main:
            link = Fn.ShortcutLink()          # dy of the block rides on conv_1's input-gradient epilogue
            x_short._grad_link = link
            x_main._add_link = link
    elif pooled_short:
        shortcut = _conv2d.Conv2D(x_short, input_dim, output_dim, 1, 1, name + '.Shortcut', spectral_normed=spectral_normed,
                                  update_collection=update_collection, he_init=False, biases=biases)
    else:
        shortcut = conv_shortcut(inputs=x_short, output_dim=output_dim, filter_size=1, name=name + '.Shortcut',
                                 spectral_normed=spectral_normed, update_collection=update_collection,
                                 he_init=False, biases=biases)

    # Normalize + nonlinearity (:185-186): relu rides on the CBN apply kernel, or on conv_1's operand
    norm1 = _normalize_kind(name + '.N1', labels) is not None
    norm2 = _normalize_kind(name + '.N2', labels) is not None
    output = Normalize(name + '.N1', x_main, labels=labels, groups=groups, relu=True)
    output = conv_1(inputs=output, filter_size=filter_size, name=name + '.Conv1',
                    spectral_normed=spectral_normed, update_collection=update_collection,
                    he_init=True, biases=biases, in_relu=not norm1, stats_groups=groups if norm2 else 0)

    fused = _norm2_inside_conv2(name, output, labels, groups, output_dim, resample, shortcut, biases, spectral_normed, out_stats) \
        if norm2 else None
    if fused is not None:
        return fused
    output = Normalize(name + '.N2', output, labels=labels, groups=groups, relu=True)
    # shortcut + output (:209) rides on conv_2's epilogue
    return conv_2(inputs=output, filter_size=filter_size, name=name + '.Conv2',
                  spectral_normed=spectral_normed, update_collection=update_collection,
                  he_init=True, biases=biases, in_relu=not norm2, residual=shortcut, stats_groups=out_stats)


FUSE_NORM_INTO_CONV2 = True   # no-grad passes: N2 + relu of an 'up' block applied while conv_2's image-resident kernel stages its operand


def _norm2_inside_conv2(name, x, labels, groups, output_dim, resample, shortcut, biases, spectral_normed, out_stats):
    """`Normalize(N2) -> nonlinearity -> conv_2 (+ shortcut)` of an 'up' block (gan_cifar_resnet.py:197-209) as ONE launch where
    nothing is kept for a backward pass and conv_2 runs on the image-resident 16x16 kernel: the 320-sample generator pass behind
    the critic updates and the sampling path (G.Block.2).  The normalised tensor (42 MB at 320 samples) is never written or read;
    same variables, same creation order (N2's tables, then Conv2's filter) as the unfused graph; bit-identical to it.  None = not
    applicable (the caller runs the unfused ops)."""
    if not (FUSE_NORM_INTO_CONV2 and resample == 'up' and not torch.is_grad_enabled() and not spectral_normed and x.is_cuda
            and _normalize_kind(name + '.N2', labels) == 'cbn' and x.dim() == 4 and tuple(x.shape[1:3]) == (16, 16)):
        return None
    n, c = x.shape[0], x.shape[3]
    if not (Fn.IMG16_CONV and K.img16_conv3x3_ok(n, (16, 16), c, output_dim) and n * (output_dim // 128) >= 256 and c == output_dim):
        return None
    store = get_default_store()
    with store.variable_scope(name + '.N2'):
        gamma, beta = _normalization.cond_batchnorm_variables(c, 10)
    filters, b = _conv2d.conv2d_variables(output_dim, output_dim, 3, 1, name + '.Conv2', he_init=True, biases=biases)
    prep = getattr(filters, '_prep_res', None)
    if prep is None or prep[0] is None:
        return None
    stats = K.cbn_stats(x, groups, getattr(x, '_cbn_stats', None))
    res_up = shortcut is not None and getattr(shortcut, '_up2x', False)
    flags = K.RES_UPSAMPLE2X if res_up else 0
    bias = b.detach() if b is not None else None
    want = out_stats if Fn.CONV_EPILOGUE_STATS else 0
    out = K.cbn_relu_img16_conv3x3(x, labels, gamma.detach(), beta.detach(), stats, prep[0], bias, output_dim, flags, shortcut, stats_groups=want)
    if want:
        y, cs = out
        y._cbn_stats = cs
        return y
    return out


FUSE_RES8 = True     # consecutive identity-shortcut 8x8x128 blocks without normalisation: one fused launch each way


def res_chain8_eligible(inputs, input_dim, names, labels=None):
    """The fused kernels (conv_resident.hip) cover blocks of 128 -> 128 channels on 8x8 images with no normalisation."""
    return (FUSE_RES8 and len(names) in (1, 2) and inputs.dim() == 4 and tuple(inputs.shape[1:]) == (8, 8, 128) and input_dim == 128
            and all(_normalize_kind(nm + '.N1', labels) is None and _normalize_kind(nm + '.N2', labels) is None for nm in names))


def ResidualBlockChain8(inputs, dim, names, spectral_normed=False, update_collection=None, biases=True, pool=False, head=None):
    """`ResidualBlock(..., resample=None)` for each name in `names`, in sequence, on 8x8 images with dim == 128 and no
    normalisation (gan_cifar_resnet.py:291-297), as one fused kernel; pool=True appends the `nonlinearity` +
    `reduce_mean(axis=[1, 2])` that follows the last critic block (:299-301).  Owns exactly the variables the
    per-block path creates (`<name>.Conv1/Filters`, ...).  head: the critic's last dense layer + hinge loss inside the two
    launches (functional.HingeHeadSpec; returns the loss)."""
    params = []
    for nm in names:
        w1, b1 = _conv2d.conv2d_variables(dim, dim, 3, 1, nm + '.Conv1', spectral_normed=spectral_normed,
                                          update_collection=update_collection, he_init=True, biases=biases)
        w2, b2 = _conv2d.conv2d_variables(dim, dim, 3, 1, nm + '.Conv2', spectral_normed=spectral_normed,
                                          update_collection=update_collection, he_init=True, biases=biases)
        params.append((w1, b1, w2, b2))
    if head is not None:
        # head = (HingeHeadSpec, variables): `variables()` creates / fetches the dense layer's (W, b) AFTER the blocks' own
        # variables, the order of the unfused graph (gan_cifar_resnet.py:291-304)
        spec, variables = head
        w_out, b_out = variables()
        return Fn.res_chain8(inputs, params, pool=True, head=(spec, w_out, b_out))
    return Fn.res_chain8(inputs, params, pool=pool)


def OptimizedResBlockDisc1(inputs, spectral_normed=False, update_collection=None, inputs_norm=False, biases=True):
    """First critic block, no pre-activation on the image (gan_cifar_resnet.py:212-234)."""
    conv_1 = functools.partial(_conv2d.Conv2D, input_dim=inputs.shape[-1], output_dim=DIM_D)
    conv_2 = functools.partial(ConvMeanPool, output_dim=DIM_D)
    conv_shortcut = MeanPoolConv
    if FUSE_FORK_POOL and FUSE_POOL_GATHER and Fn.fork_pool_conv1x1_ok(inputs, DIM_D):
        # the 2x2 mean inside the 1x1 conv's gather: fork + pool + conv as one launch
        w_s, b_s = _conv2d.conv2d_variables(inputs.shape[-1], DIM_D, 1, 1, 'D.Block.1.Shortcut', spectral_normed=spectral_normed,
                                            update_collection=update_collection, he_init=False, biases=biases)
        w_1, b_1 = _conv2d.conv2d_variables(inputs.shape[-1], DIM_D, 3, 1, 'D.Block.1.Conv1', spectral_normed=spectral_normed,
                                            update_collection=update_collection, he_init=True, biases=biases)
        x_main, shortcut = Fn.fork_pool_conv1x1(inputs, w_s, b_s, w_1, b_1)      # ... and conv_1 below on the same launch
    elif FUSE_FORK_POOL:
        x_main, pooled = Fn.fork_pool(inputs)
        shortcut = _conv2d.Conv2D(pooled, pooled.shape[-1], DIM_D, 1, 1, 'D.Block.1.Shortcut', spectral_normed=spectral_normed,
                                  update_collection=update_collection, he_init=False, biases=biases)
    else:
        x_short, x_main = Fn.fork(inputs)
        shortcut = conv_shortcut(inputs=x_short, output_dim=DIM_D, filter_size=1, name='D.Block.1.Shortcut',
                                 spectral_normed=spectral_normed, update_collection=update_collection,
                                 he_init=False, biases=biases)
    output = conv_1(inputs=x_main, filter_size=3, name='D.Block.1.Conv1',
                    spectral_normed=spectral_normed, update_collection=update_collection,
                    he_init=True, biases=biases)
    return conv_2(inputs=output, filter_size=3, name='D.Block.1.Conv2',
                  spectral_normed=spectral_normed, update_collection=update_collection,
                  he_init=True, biases=biases, in_relu=True, residual=shortcut)
