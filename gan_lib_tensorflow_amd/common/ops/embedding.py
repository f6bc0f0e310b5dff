"""Label embedding -- drop-in for common/ops/embedding.py of the reference."""
import numpy as np

from ... import functional as Fn
from ...store import get_default_store


def embed_y(inputs, vocab_size=1000, embedding_dim=300, word2vec_file=None,
            spectral_normed=False, update_collection=None, reuse=False):
    """inputs: int32 [batch]; returns bf16 [batch, embedding_dim] (embedding.py:12-51)."""
    store = get_default_store()
    with store.variable_scope("Embedding.Label"):
        if word2vec_file is None:
            table = store.get_variable('embedding_map', [vocab_size, embedding_dim],
                                       lambda rng: rng.uniform(low=-0.08, high=0.08,
                                                               size=(vocab_size, embedding_dim)).astype('float32'),
                                       trainable=True)
        else:
            table = store.get_variable('embedding_map', None, np.asarray(word2vec_file, 'float32'), trainable=False)
        return Fn.embedding(table, inputs)
