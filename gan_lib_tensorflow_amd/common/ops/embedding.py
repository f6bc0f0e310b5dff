"""Label embedding -- drop-in for common/ops/embedding.py of the reference."""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store


def embedding_variable(vocab_size=1000, embedding_dim=300, word2vec_file=None):
    """The variable half of embed_y (embedding.py:28-40): `Embedding.Label/embedding_map` [vocab, dim], U(+-0.08) or the
    word2vec table (then not trainable)."""
    store = get_default_store()
    with store.variable_scope("Embedding.Label"):
        if word2vec_file is None:
            return store.get_variable('embedding_map', [vocab_size, embedding_dim],
                                      lambda rng: rng.uniform(low=-0.08, high=0.08,
                                                              size=(vocab_size, embedding_dim)).astype('float32'),
                                      trainable=True)
        return store.get_variable('embedding_map', None, np.asarray(word2vec_file, 'float32'), trainable=False)


def embed_y(inputs, vocab_size=1000, embedding_dim=300, word2vec_file=None,
            spectral_normed=False, update_collection=None, reuse=False):
    """inputs: int32 [batch]; returns bf16 [batch, embedding_dim] (embedding.py:12-51)."""
    return Fn.embedding(embedding_variable(vocab_size, embedding_dim, word2vec_file), inputs)
