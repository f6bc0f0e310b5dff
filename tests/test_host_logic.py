"""CPU: host-side logic of the drop-in layer -- variable naming/scopes, flat buffers, SN name
pairing, LR schedule, synthetic feed, argument errors mirrored from the reference."""
import numpy as np
import pytest
import torch

from gan_lib_tensorflow_amd.store import ParamStore, set_default_store
from gan_lib_tensorflow_amd.common.ops import sn as sn_mod
from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
from oracle import ref_ops as R


def test_scopes_names_and_get_or_create():
    st = ParamStore("cpu", seed=0)
    with st.variable_scope("Discriminator"):
        with st.variable_scope("D.Block.1.Conv1"):
            w = st.get_variable("Filters", [3, 3, 3, 8], lambda rng: rng.uniform(size=(3, 3, 3, 8)))
            with st.variable_scope("filters"), st.variable_scope("spectral_norm"):
                u = st.get_variable("u", [1, 8], np.ones((1, 8)), trainable=False)
            w2 = st.get_variable("Filters")
    assert w2 is w and w.requires_grad and not u.requires_grad
    assert list(st.vars) == ["Discriminator/D.Block.1.Conv1/Filters", "Discriminator/D.Block.1.Conv1/filters/spectral_norm/u"]
    assert st.params_with_name("Discriminator") == [w]
    with pytest.raises(KeyError):
        st.get_variable("missing")
    with pytest.raises(ValueError):
        st.get_variable("bad", [2], np.zeros(3))
    pairs = sn_mod.sn_pairs(st, "Discriminator")
    assert len(pairs) == 1 and pairs[0][0] is w and pairs[0][1] is u


def test_flatten_views_alignment_and_state_dict_roundtrip():
    st = ParamStore("cpu", seed=1)
    with st.variable_scope("Generator"):
        a = st.get_variable("a", [3], np.arange(3.))
        b = st.get_variable("b", [2, 5], np.arange(10.).reshape(2, 5))
        st.get_variable("u", [1, 4], np.ones((1, 4)), trainable=False)
    flat = st.flatten("Generator")
    assert flat["params"].numel() == 4 + 12 and flat["offsets"] == {"Generator/a": 0, "Generator/b": 4}
    assert a.data_ptr() == flat["params"].data_ptr() and b.data_ptr() == flat["params"].data_ptr() + 16
    assert a.main_grad.data_ptr() == flat["grads"].data_ptr()
    b.main_grad += 1
    assert float(flat["grads"].sum()) == 10.
    st.zero_grads("Generator")
    assert float(flat["grads"].abs().sum()) == 0.
    with torch.no_grad():
        flat["params"].mul_(2)
    assert float(b[1, 4]) == 18.
    sd = st.state_dict()
    st2 = ParamStore("cpu")
    with st2.variable_scope("Generator"):
        st2.get_variable("a", [3], np.zeros(3))
        st2.get_variable("b", [2, 5], np.zeros((2, 5)))
        st2.get_variable("u", [1, 4], np.zeros((1, 4)), trainable=False)
    st2.load_state_dict(sd)
    assert float(st2.vars["Generator/b"][1, 4]) == 18.
    with pytest.raises(RuntimeError):
        with st.variable_scope("Generator"):
            st.get_variable("late", [1], np.zeros(1))


def test_lr_decay_matches_reference_schedule():
    for it in (0, 1, 25000, 49999, 50000, 99999):
        assert S.lr_decay(it) == R.lr_decay(it)


def test_reference_error_behaviour_is_mirrored():
    from gan_lib_tensorflow_amd.common import resnet_block as B
    from gan_lib_tensorflow_amd.common.ops import conv2d, normalization
    set_default_store(ParamStore("cpu"))
    x = torch.zeros(1, 4, 4, 8, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        conv2d.Conv2D(x, 8, 8, conv_type="depthwise_conv2d", name="c")        # conv2d.py:210
    with pytest.raises(Exception, match="invalid resample value"):
        B.ResidualBlock(x, 8, 8, 3, "D.x", resample="sideways")               # gan_cifar_resnet.py:174
    with pytest.raises(Exception, match="Axes is not supported"):
        normalization.cond_batchnorm("n", [0, 1], x, labels=None, n_labels=10)  # normalization.py:45


def test_synthetic_feed_shapes():
    it = S.synthetic_batches(4, "cpu", seed=0)
    d, l = next(it)
    assert d.shape == (4, 3072) and d.dtype == torch.uint8 and l.dtype == torch.int32 and int(l.max()) <= 9


def test_sampling_quantisation_and_inception_score_statistics():
    """Host side of the sampling / IS harness (gan_cifar_resnet.py:536-551, inception_score.py:59-90): known answers."""
    from gan_lib_tensorflow_amd.common.inception.inception_score import quantize_samples, preds2score, get_inception_score
    x = np.array([-1.0, -0.5, 0.0, 0.999, 1.0], np.float32)
    np.testing.assert_array_equal(quantize_samples(x), [0, 63, 127, 254, 255])                 # (x+1)*127.5, truncated
    np.testing.assert_array_equal(quantize_samples(x, for_score=True), [0, 63, 127, 255, 255])  # (x+1)*127.995
    # every sample predicts the marginal: KL = 0 -> score 1, std 0
    uni = np.full((100, 10), 0.1)
    m, s = preds2score(uni, 10)
    assert abs(m - 1.0) < 1e-12 and s < 1e-12
    # confident and evenly spread over K classes inside every split: score -> K
    K_, eps = 5, 1e-9
    p = np.full((100, K_), eps)
    p[np.arange(100), np.arange(100) % K_] = 1.0 - (K_ - 1) * eps
    m, s = preds2score(p, 10)
    assert abs(m - K_) < 1e-5 and s < 1e-9
    # against an independent entropy formulation: exp(H(marginal) - mean H(p(y|x)))
    rng = np.random.default_rng(0)
    q = rng.dirichlet(np.ones(7), size=60)
    H = lambda r: -(r * np.log(r)).sum(-1)            # noqa: E731
    want = [np.exp(H(q[i * 20:(i + 1) * 20].mean(0)) - H(q[i * 20:(i + 1) * 20]).mean()) for i in range(3)]
    m, s = preds2score(q, 3)
    assert abs(m - np.mean(want)) < 1e-12 and abs(s - np.std(want)) < 1e-12
    # the classifier is a parameter (no Inception weights here): pixel-valued input is mapped to [-1,1], whole batches only
    seen = []
    def clf(b):
        seen.append((b.min(), b.max(), b.shape))
        return np.tile(np.arange(1008, dtype=np.float64) * 0.0, (b.shape[0], 1))
    imgs = rng.integers(0, 256, (130, 4, 4, 3))
    m, s = get_inception_score(imgs, splits=2, classifier=clf, batch_size=64)
    assert len(seen) == 2 and all(lo >= -1 and hi <= 1 and sh == (64, 4, 4, 3) for lo, hi, sh in seen) and abs(m - 1.0) < 1e-12
    with pytest.raises(NotImplementedError):
        get_inception_score(imgs)
