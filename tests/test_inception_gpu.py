"""The Inception-v3 classifier of the Inception-score harness (common/inception/inception_v3.py) on the HIP kernels against the
float64 restatement (oracle/ref_inception.py), random weights in the frozen graph's layout (the real weights are a download)."""
import os

import numpy as np
import pytest
import torch


def random_params(seed=0):
    from gan_lib_tensorflow_amd.common.inception.inception_v3 import param_shapes
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes().items():
        if name.endswith('conv2d_params'):
            fan_in = shape[0] * shape[1] * shape[2]
            p[name] = (rng.normal(size=shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)      # keeps activations O(1) through 47 layers
        elif name.endswith('moving_variance'):
            p[name] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif name.endswith(('beta', 'moving_mean')):
            p[name] = (0.1 * rng.normal(size=shape)).astype(np.float32)
        elif name == 'softmax/weights':
            p[name] = (rng.normal(size=shape) / np.sqrt(2048)).astype(np.float32)
        else:
            p[name] = (0.1 * rng.normal(size=shape)).astype(np.float32)
    return p


def test_weights_file_layout_is_the_2015_graph():
    """CPU: the parameter list has the scopes, shapes and size of the published network (23.8 M filter weights + the 1008-way head)."""
    from gan_lib_tensorflow_amd.common.inception.inception_v3 import conv_layers, param_shapes
    layers = conv_layers()
    assert len(layers) == 94
    shapes = param_shapes()
    assert shapes['conv/conv2d_params'] == (3, 3, 3, 32) and shapes['mixed_4/tower/conv_1/conv2d_params'] == (1, 7, 128, 128)
    assert shapes['mixed_10/tower_1/mixed/conv_1/conv2d_params'] == (3, 1, 384, 384) and shapes['softmax/weights'] == (2048, 1008)
    n = sum(int(np.prod(s)) for k, s in shapes.items() if k.endswith('conv2d_params'))
    assert n == 21751136 and n + 2048 * 1008 == 23815520, n      # "23.8 M parameters" with the 1008-way head
    # every block's branches add up to the next block's input width
    cin = {nm: ci for nm, _, _, ci, _ in layers}
    assert cin['mixed_1/conv'] == 256 and cin['mixed_2/conv'] == 288 and cin['mixed_3/conv'] == 288 and cin['mixed_4/conv'] == 768
    assert cin['mixed_8/tower/conv'] == 768 and cin['mixed_9/conv'] == 1280 and cin['mixed_10/conv'] == 2048


def test_oracle_pieces_known_answers():
    """CPU: the restatement's TF-1 bilinear resize on an index ramp, and SAME average pooling that does not count the padding."""
    from oracle import ref_inception as RI
    x = torch.arange(16, dtype=torch.float64).reshape(1, 1, 4, 4)
    y = RI.resize_bilinear_tf1(x, 8)
    assert float(y[0, 0, 0, 1]) == 0.5 and float(y[0, 0, 1, 0]) == 2.0 and float(y[0, 0, 7, 7]) == 15.0        # clamped at the far edge
    ones = torch.ones(1, 1, 5, 5, dtype=torch.float64)
    assert float((RI.Net.avg3(ones) - 1).abs().max()) == 0.0


@pytest.mark.gpu
def test_pool2d_and_relu_to_channels():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.nn.functional as F
    from gan_lib_tensorflow_amd import kernels as K
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 9, 9, 16, generator=g).to(torch.bfloat16)
    xc = x.double().permute(0, 3, 1, 2)
    for k, s, pad, mode, ref in ((3, 2, 0, 'max', F.max_pool2d(xc, 3, 2)), (3, 1, 1, 'max', F.max_pool2d(xc, 3, 1, 1)),
                                 (3, 1, 1, 'avg', F.avg_pool2d(xc, 3, 1, 1, count_include_pad=False)), (9, 1, 0, 'avg', xc.mean((2, 3), keepdim=True))):
        out_hw = tuple(ref.shape[2:])
        y = K.pool2d(x.cuda(), k, s, pad, out_hw, mode)
        assert float((y.double().cpu().permute(0, 3, 1, 2) - ref).abs().max()) < 2e-2, (k, s, pad, mode)
    wide = torch.zeros(2, 9, 9, 40, dtype=torch.bfloat16, device="cuda")
    K.relu_to_channels(x.cuda(), wide, 8)
    K.pool2d(x.cuda(), 3, 1, 1, (9, 9), 'max', out=wide, c_off=24)
    assert torch.equal(wide[..., 8:24].cpu(), torch.relu(x)) and float(wide[..., :8].abs().max()) == 0.0
    assert float((wide[..., 24:].double().cpu().permute(0, 3, 1, 2) - F.max_pool2d(xc, 3, 1, 1)).abs().max()) == 0.0


@pytest.mark.gpu
def test_inception_v3_forward_matches_the_float64_restatement():
    """Random weights in the frozen graph's layout, 32 x 32 inputs in [-1, 1] resized to 299 x 299 (the harness's path for CIFAR
    samples): pool_3 features and logits of the HIP network against float64.  Stated tolerance for 47 convolutions deep in
    bf16 storage: relative L2 <= 0.02 and cosine >= 0.9995 on features and logits (measured 0.004), the same top-1 class."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gan_lib_tensorflow_amd.common.inception.inception_v3 import InceptionV3
    from gan_lib_tensorflow_amd.common.inception.inception_score import get_inception_score
    from oracle import ref_inception as RI
    torch.set_num_threads(16)
    params = random_params(3)
    rng = np.random.default_rng(1)
    images = torch.tensor(rng.uniform(-1, 1, size=(2, 32, 32, 3)).astype(np.float32)).to(torch.bfloat16)
    net = InceptionV3(params)
    feat = net.features(images).double().cpu()
    logits = torch.tensor(net.logits(images)).double()
    ref = RI.Net(params)
    rfeat = ref.features(images.double())
    rlog = rfeat @ ref.p['softmax/weights'] + ref.p['softmax/biases']
    for name, got, want in (("pool_3", feat, rfeat), ("logits", logits, rlog)):
        l2 = float((got - want).norm() / want.norm())
        cos = float((got * want).sum() / (got.norm() * want.norm()))
        print(name, "relative L2", round(l2, 4), "cosine", round(cos, 5))
        assert l2 < 0.02 and cos > 0.9995, (name, l2, cos)          # measured 0.004 / 0.99999
    assert torch.equal(logits[:, :1000].argmax(1), rlog[:, :1000].argmax(1))
    # the scoring loop end to end on this classifier: whole batches, first 1000 logits, softmax, 10-split KL score
    many = rng.uniform(-1, 1, size=(40, 32, 32, 3)).astype(np.float32)
    mean, std = get_inception_score(many, splits=2, classifier=net.logits, batch_size=20)
    assert np.isfinite(mean) and mean >= 1.0 and np.isfinite(std)


@pytest.mark.gpu
def test_inception_score_protocol_end_to_end_vs_the_float64_restatement():
    """SNGAN/gan_cifar_resnet.py:543-555 wired through `SNGANTrainer.inception_score`: n / 100 generator passes of 100 samples on
    fresh uniform labels, `(x + 1) * 255.99 / 2` -> int32 on the host, back to [-1, 1], whole classifier batches, first 1000 logits,
    softmax, 10-split exp(KL).  The HIP Inception network (random weights in the frozen graph's layout) classifies 1000 samples;
    the SAME quantised samples go through the float64 restatement (oracle/ref_inception.py) in the same batches, and the two scores
    agree to 2 % (bf16 storage 47 convolutions deep moves the logits by 0.4 %; the score is a smooth function of them).  Also the
    every-1000-iterations hook (:634-637)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gan_lib_tensorflow_amd.SNGAN import gan_cifar_resnet as S
    from gan_lib_tensorflow_amd.common.inception.inception_v3 import InceptionV3
    from gan_lib_tensorflow_amd.common.inception import inception_score as IS
    from oracle import ref_inception as RI
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    params = random_params(5)
    net = InceptionV3(params)
    ref = RI.Net(params)
    tr = S.SNGANTrainer(batch_size=8, seed=4, use_graphs=False)
    seen = []

    def classifier(batch):                      # the HIP network; keeps what it was shown for the restatement
        seen.append(np.array(batch, copy=True))
        return net.logits(torch.tensor(batch).to(torch.bfloat16))
    n = 1000
    mean, std = tr.inception_score(n=n, classifier=classifier, splits=10, batch_size=100)
    assert len(seen) == n // 100 and all(b.shape == (100, 32, 32, 3) for b in seen)
    shown = np.concatenate(seen, 0)
    # what the classifier saw is the quantised sample grid: k / 127.5 - 1 for integers k in [0, 255]
    k = (shown + 1.0) * 127.5
    assert np.abs(k - np.round(k)).max() < 1e-3 and k.min() >= -1e-3 and k.max() <= 255 + 1e-3
    assert np.isfinite(mean) and mean >= 1.0 and np.isfinite(std)

    def ref_classifier(batch):
        x = torch.tensor(batch).to(torch.bfloat16).double()      # the HIP network receives bf16 images
        f = ref.features(x)
        return (f @ ref.p['softmax/weights'] + ref.p['softmax/biases']).numpy()
    sub = 200                                   # the float64 network on 200 of the 1000 samples (a minute of CPU), same protocol, 2 splits
    m_hip, _ = IS.get_inception_score(shown[:sub], splits=2, classifier=lambda b: net.logits(torch.tensor(b).to(torch.bfloat16)), batch_size=100)
    m_ref, _ = IS.get_inception_score(shown[:sub], splits=2, classifier=ref_classifier, batch_size=100)
    print("inception score (random weights): HIP", m_hip, "float64", m_ref, "| 1000 samples, 10 splits:", mean, std)
    assert abs(m_hip - m_ref) < 0.02 * m_ref, (m_hip, m_ref)
    # the hook: due every INCEPTION_FREQUENCY finished iterations
    tr.iteration = S.INCEPTION_FREQUENCY - 1
    assert tr.maybe_inception_score(classifier, n=100) is None
    tr.iteration = S.INCEPTION_FREQUENCY
    got = tr.maybe_inception_score(classifier, n=100)
    assert got is not None and np.isfinite(got[0])
