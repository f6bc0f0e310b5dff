"""Normalisation layers -- drop-in for common/ops/normalization.py of the reference (hot path:
`cond_batchnorm`; the unconditional / layer / instance / pixel variants belong to configs 3-5)."""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store


def cond_batchnorm_variables(c, n_labels):
    """The variable half of cond_batchnorm (normalization.py:43,49-51): `CondBatchNorm/{offset,scale}` [n_labels, c] under
    the current scope; returns (scale, offset).  Shared with kernels that normalise inside a consumer."""
    store = get_default_store()
    with store.variable_scope('CondBatchNorm'):
        offset_m = store.get_variable('offset', [n_labels, c], np.zeros((n_labels, c), 'float32'))
        scale_m = store.get_variable('scale', [n_labels, c], np.ones((n_labels, c), 'float32'))
    return scale_m, offset_m


def cond_batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
                   labels=None, n_labels=None, groups=1, relu=False):
    """Conditional Batchnorm (dumoulin et al 2016) for BHWC conv filtermaps (normalization.py:27-59).
    Batch statistics always (there are no moving averages in the reference).  `groups`: number of
    towers with independent statistics; `relu`: fuse the following nonlinearity."""
    store = get_default_store()
    with store.variable_scope('CondBatchNorm'):
        if axes != [0, 1, 2]:
            raise Exception('Axes is not supported in Conditional BatchNorm!')
        c = inputs.shape[3]
        offset_m = store.get_variable('offset', [n_labels, c], np.zeros((n_labels, c), 'float32'))
        scale_m = store.get_variable('scale', [n_labels, c], np.ones((n_labels, c), 'float32'))
        return Fn.cond_batchnorm(inputs, labels, scale_m, offset_m, groups, relu)


_zero_labels = {}


def batch_norm(inputs, decay=0.9, epsilon=1e-5, is_training=True, fused=True, groups=1, relu=False):
    """tf.contrib.layers.batch_norm(center, scale, updates_collections=None, zero_debias_moving_mean=True, fused,
    scope='BatchNorm')  (normalization.py:8-24): train-mode batch normalisation over (N,H,W) with in-place
    moving-statistic updates.  Variables `BatchNorm/{beta,gamma,moving_mean,moving_variance}` and the zero-debias
    helpers `BatchNorm/moving_mean/{biased,local_step}`.  Runs on the conditional-batch-norm kernels with a
    one-row gamma/beta table; the moving statistics (state only: every call site of the reference trains) are
    updated per tower like the reference's per-tower update ops:
        moving_variance <- decay*mv + (1-decay) * var * n/(n-1)      (fused batch norm reports the unbiased variance)
        biased <- decay*biased + (1-decay)*mean; local_step += 1; moving_mean <- biased / (1 - decay**local_step)"""
    if not is_training:
        raise NotImplementedError('the reference only ever calls batch_norm with is_training=True (normalization.py:8)')
    if abs(epsilon - 1e-5) > 1e-12:
        raise NotImplementedError('the batch-norm kernels are built with epsilon = 1e-5 (the reference default)')
    import torch
    store = get_default_store()
    n, c = inputs.shape[0], inputs.shape[-1]
    with store.variable_scope('BatchNorm'):
        beta = store.get_variable('beta', [1, c], np.zeros((1, c), 'float32'))
        gamma = store.get_variable('gamma', [1, c], np.ones((1, c), 'float32'))
        mm = store.get_variable('moving_mean', [c], np.zeros(c, 'float32'), trainable=False)
        mv = store.get_variable('moving_variance', [c], np.ones(c, 'float32'), trainable=False)
        with store.variable_scope('moving_mean'):
            biased = store.get_variable('biased', [c], np.zeros(c, 'float32'), trainable=False)
            step = store.get_variable('local_step', [1], np.zeros(1, 'float32'), trainable=False)
    key = (n, str(inputs.device))
    if key not in _zero_labels:
        _zero_labels[key] = torch.zeros(n, dtype=torch.int32, device=inputs.device)
    y, stats = Fn.batchnorm_with_stats(inputs, _zero_labels[key], gamma, beta, groups, relu)
    from ... import kernels as K
    K.bn_moving_update(stats, mm, mv, biased, step, inputs.numel() // c // groups, decay)     # one launch, towers in order
    return y


def layer_norm(name, norm_axes, inputs):
    """tf.contrib.layers.layer_norm(center, scale, begin_norm_axis=1, begin_params_axis=-1, scope=name)
    (normalization.py:62-102): moments over every non-batch axis per sample (variance epsilon 1e-12, TF's constant),
    `gamma` / `beta` over the last axis, created under scope `name`."""
    store = get_default_store()
    c = inputs.shape[-1]
    with store.variable_scope(name):
        beta = store.get_variable('beta', [c], np.zeros(c, 'float32'))
        gamma = store.get_variable('gamma', [c], np.ones(c, 'float32'))
    return Fn.layer_norm(inputs, gamma, beta, 1e-12)


def instance_norm(inputs, epsilon=1e-06):
    """tf.contrib.layers.instance_norm(center, scale, data_format='NHWC')  (normalization.py:105-122): moments over
    (H,W) per sample and channel -- the conditional-batch-norm kernels with one tower per sample and a one-row table."""
    import torch
    store = get_default_store()
    n, c = inputs.shape[0], inputs.shape[-1]
    with store.variable_scope('InstanceNorm'):
        beta = store.get_variable('beta', [1, c], np.zeros((1, c), 'float32'))
        gamma = store.get_variable('gamma', [1, c], np.ones((1, c), 'float32'))
    key = (n, str(inputs.device))
    if key not in _zero_labels:
        _zero_labels[key] = torch.zeros(n, dtype=torch.int32, device=inputs.device)
    y, _ = Fn.batchnorm_with_stats(inputs, _zero_labels[key], gamma, beta, groups=n, relu=False, eps=epsilon)
    return y


def pixel_norm(inputs, eps=1e-8):
    """From PGGAN (normalization.py:125-140): inputs * rsqrt(mean(inputs**2, axis=3) + eps)."""
    return Fn.pixel_norm(inputs, eps)
