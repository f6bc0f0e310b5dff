"""Normalisation layers -- drop-in for common/ops/normalization.py of the reference (hot path:
`cond_batchnorm`; the unconditional / layer / instance / pixel variants belong to configs 3-5)."""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store


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


def batch_norm(inputs, decay=0.9, epsilon=1e-5, is_training=True, fused=True):
    raise NotImplementedError('unconditional batch_norm (normalization.py:8-24) is config 3 (ACGAN), not built yet')


def layer_norm(name, norm_axes, inputs):
    raise NotImplementedError('layer_norm (normalization.py:62-102) is only reached with NORMALIZATION_D=True')


def instance_norm(inputs, epsilon=1e-06):
    raise NotImplementedError('instance_norm (normalization.py:105-122) belongs to Pix2Pix (config 5)')


def pixel_norm(inputs, eps=1e-8):
    raise NotImplementedError('pixel_norm (normalization.py:125-140) belongs to PGGAN (config 4)')
