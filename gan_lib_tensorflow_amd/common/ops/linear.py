"""Dense layer -- drop-in for common/ops/linear.py of the reference.

Variables `<name>/W` [in,out], `<name>/b`, `<name>/spectral_norm/u`; default initialisation is the
Glorot-uniform branch (linear.py:76-80; `initialization=None` never reaches the orthogonal branch).
tf.matmul + bias_add (linear.py:161-180) run as the 1x1 case of the MFMA conv engine.
"""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store
from .sn import spectral_normed_weight

_default_weightnorm = False
_weights_stdev = None


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


def disable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = False


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def linear_variables(input_dim, output_dim, name, spectral_normed=False, update_collection=None, inputs_norm=False,
                     biases=True, initialization=None, weightnorm=None, gain=1.):
    """The variable half of Linear (linear.py:45-177): creates / fetches `<name>/W` (spectrally normalised when asked) and
    `<name>/b`; returns (weight, biases | None).  Shared by Linear and by fused heads, so both own identical variables."""
    store = get_default_store()
    with store.variable_scope(name):
        if inputs_norm or (weightnorm if weightnorm is not None else _default_weightnorm):
            raise NotImplementedError('inputs_norm / weightnorm are outside the SNGAN hot path')

        def uniform(rng, stdev, size):
            if _weights_stdev is not None:
                stdev = _weights_stdev
            return rng.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')

        def init(rng):
            if initialization == 'lecun':
                w = uniform(rng, np.sqrt(1. / input_dim), (input_dim, output_dim))
            elif initialization == 'glorot' or initialization == 'xavier' or (initialization is None):
                w = uniform(rng, np.sqrt(2. / (input_dim + output_dim)), (input_dim, output_dim))
            elif initialization == 'he':
                w = uniform(rng, np.sqrt(2. / input_dim), (input_dim, output_dim))
            elif initialization == 'glorot_he':
                w = uniform(rng, np.sqrt(4. / (input_dim + output_dim)), (input_dim, output_dim))
            elif initialization == 'orthogonal':
                a = rng.normal(0.0, 1.0, (input_dim, output_dim))
                u, _, v = np.linalg.svd(a, full_matrices=False)
                w = (u if u.shape == (input_dim, output_dim) else v).astype('float32')
            elif initialization[0] == 'uniform':
                w = rng.uniform(low=-initialization[1], high=initialization[1],
                                size=(input_dim, output_dim)).astype('float32')
            else:
                raise Exception('Invalid initialization!')
            return w * gain

        weight = store.get_variable('W', [input_dim, output_dim], init)
        if spectral_normed:
            weight = spectral_normed_weight(weight, update_collection=update_collection)
        _biases = None
        if biases:
            _biases = store.get_variable('b', [output_dim], np.zeros(output_dim, 'float32'))
        return weight, _biases


def Linear(inputs, input_dim, output_dim, name,
           spectral_normed=False, update_collection=None, reuse=False, inputs_norm=False,
           biases=True, initialization=None, weightnorm=None, gain=1.):
    """initialization: None, `lecun`, 'glorot', `he`, 'glorot_he', `("uniform", range)`"""
    weight, _biases = linear_variables(input_dim, output_dim, name, spectral_normed, update_collection, inputs_norm, biases,
                                       initialization, weightnorm, gain)
    lead = inputs.shape[:-1]
    result = Fn.linear(inputs.reshape(-1, input_dim), weight, _biases)
    return result.reshape(*lead, output_dim)
