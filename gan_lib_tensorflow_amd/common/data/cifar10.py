"""CIFAR-10 input pipeline -- drop-in for common/data/cifar10.py of the reference.

Same entry points and file format: the python-version pickles of the CIFAR-10 distribution (`data_batch_1` .. `_5`,
`test_batch`), each a dict with b'data' (uint8 [10000, 3072], rows CHW-planar: 1024 red, 1024 green, 1024 blue bytes) and
b'labels' (list of ints); `load(batch_size, data_dir)` returns two epoch-generator factories (train, dev) that yield
(uint8 [B, 3072], labels [B]) and drop the last partial batch (cifar10.py:18-45).  The rows go to the critic as they are:
`gank_preprocess_real` (kernels.preprocess_real / the trainer's feed ring) does the int -> float, /256, the dequantisation noise
and the CHW -> HWC transpose of SNGAN/gan_cifar_resnet.py:334-338 on the device.

The epoch order is the reference's for a given `np.random` state: it shuffles the image array and the label array in place with
the same generator state; here ONE index array is shuffled with that state and the order accumulates from epoch to epoch, which
visits the same rows in the same order without moving 150 MB per epoch.
"""
import os
import pickle

import numpy as np


def unpickle(file):
    """-> (data uint8 [n, 3072], labels list[int])   (cifar10.py:9-15)"""
    with open(file, 'rb') as fo:
        d = pickle.load(fo, encoding='bytes')
    return d[b'data'], d[b'labels']


def cifar_generator(filenames, batch_size, data_dir):
    """-> get_epoch: a callable returning a generator over one shuffled epoch of whole batches (cifar10.py:18-37)"""
    all_data, all_labels = [], []
    for filename in filenames:
        data, labels = unpickle(os.path.join(data_dir, filename))
        all_data.append(np.asarray(data, dtype=np.uint8))
        all_labels.append(np.asarray(labels))
    images = np.ascontiguousarray(np.concatenate(all_data, axis=0))
    labels = np.concatenate(all_labels, axis=0)
    if images.ndim != 2 or images.shape[1] != 3072 or len(labels) != len(images):
        raise ValueError(f"expected uint8 rows of 3072 bytes and one label per row, got {images.shape} / {labels.shape}")
    order = np.arange(len(images))

    def get_epoch():
        nonlocal order
        # np.random.shuffle draws the same swaps for any array of this length: shuffling the current order with the global
        # state is the permutation the reference applies to its (already shuffled) image and label arrays
        idx = np.arange(len(order))
        np.random.shuffle(idx)
        order = order[idx]
        for i in range(len(images) // batch_size):
            sel = order[i * batch_size:(i + 1) * batch_size]
            yield images[sel], labels[sel]

    return get_epoch


def load(batch_size, data_dir):
    """-> (train epoch factory, dev epoch factory)   (cifar10.py:40-45)"""
    return (cifar_generator(['data_batch_1', 'data_batch_2', 'data_batch_3', 'data_batch_4', 'data_batch_5'], batch_size, data_dir),
            cifar_generator(['test_batch'], batch_size, data_dir))


def inf_train_gen(train_gen):
    """the endless batch stream of SNGAN/gan_cifar_resnet.py:560-566"""
    while True:
        for images_, labels_ in train_gen():
            yield images_, labels_


def device_batches(gen, device):
    """(uint8 [B,3072], labels) numpy pairs -> the tensors SNGANTrainer.train_iteration / d_step take (uint8 rows and int32
    labels on `device`); pinned staging buffers, non-blocking copies."""
    import torch
    for images_, labels_ in gen:
        x = torch.from_numpy(np.ascontiguousarray(images_))
        y = torch.from_numpy(np.ascontiguousarray(labels_).astype(np.int32))
        if torch.device(device).type == 'cuda':
            x, y = x.pin_memory(), y.pin_memory()
        yield x.to(device, non_blocking=True), y.to(device, non_blocking=True)
