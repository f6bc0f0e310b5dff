"""common/data/cifar10.py mirror on a synthetic pickle set in the CIFAR-10 layout (the real files are a download)."""
import os
import pickle

import numpy as np

from gan_lib_tensorflow_amd.common.data import cifar10


def _write_set(d, rows_per_file=40, seed=0):
    rng = np.random.default_rng(seed)
    names = ['data_batch_1', 'data_batch_2', 'data_batch_3', 'data_batch_4', 'data_batch_5', 'test_batch']
    allx, ally = {}, {}
    for i, nm in enumerate(names):
        x = rng.integers(0, 256, (rows_per_file, 3072), dtype=np.uint8)
        x[:, 0] = np.arange(rows_per_file) + 40 * (i % 5)       # a row id in byte 0 (train files: 0..199)
        y = [int(v) for v in rng.integers(0, 10, rows_per_file)]
        with open(os.path.join(d, nm), 'wb') as f:
            pickle.dump({b'data': x, b'labels': y, b'batch_label': b'synthetic', b'filenames': []}, f)
        allx[nm], ally[nm] = x, np.asarray(y)
    return names, allx, ally


def test_epochs_are_shuffled_whole_batches_with_matching_labels(tmp_path):
    names, allx, ally = _write_set(str(tmp_path))
    train, dev = cifar10.load(64, str(tmp_path))
    x = np.concatenate([allx[n] for n in names[:5]])
    y = np.concatenate([ally[n] for n in names[:5]])
    np.random.seed(7)
    seen = []
    for ep in range(2):
        batches = list(train())
        assert len(batches) == 200 // 64                       # the last partial batch is dropped (cifar10.py:34)
        ids = np.concatenate([b[0][:, 0] for b in batches])
        assert len(set(ids.tolist())) == len(ids)              # no row twice in an epoch
        for bx, by in batches:
            assert bx.dtype == np.uint8 and bx.shape == (64, 3072) and by.shape == (64,)
            assert np.array_equal(bx, x[bx[:, 0]]) and np.array_equal(by, y[bx[:, 0]])      # labels travel with their rows
        seen.append(ids)
    assert not np.array_equal(seen[0], seen[1])                # a new order every epoch
    assert len(list(dev())) == 0 and len(list(cifar10.load(8, str(tmp_path))[1]())) == 5   # 40 dev rows


def test_epoch_order_is_the_in_place_shuffle_of_the_reference(tmp_path):
    """The reference shuffles images and labels IN PLACE with the same np.random state, epoch after epoch (cifar10.py:29-32)."""
    names, allx, ally = _write_set(str(tmp_path), seed=1)
    x = np.concatenate([allx[n] for n in names[:5]])
    np.random.seed(3)
    want = []
    for ep in range(3):
        st = np.random.get_state()
        np.random.shuffle(x)
        np.random.set_state(st)
        np.random.shuffle(np.arange(len(x)))                   # the labels' shuffle consumes the same draws
        want.append(x[:192, 0].copy())
    train, _ = cifar10.load(64, str(tmp_path))
    np.random.seed(3)
    for ep in range(3):
        st = np.random.get_state()
        got = np.concatenate([b[0][:, 0] for b in train()])
        np.random.set_state(st)
        np.random.shuffle(np.arange(200))
        assert np.array_equal(got, want[ep]), ep


def test_inf_train_gen_and_device_batches(tmp_path):
    import torch
    _write_set(str(tmp_path))
    train, _ = cifar10.load(32, str(tmp_path))
    it = cifar10.device_batches(cifar10.inf_train_gen(train), 'cpu')
    for _ in range(15):                                         # more than two epochs of 6 batches
        xb, yb = next(it)
        assert xb.dtype == torch.uint8 and tuple(xb.shape) == (32, 3072) and yb.dtype == torch.int32 and tuple(yb.shape) == (32,)
