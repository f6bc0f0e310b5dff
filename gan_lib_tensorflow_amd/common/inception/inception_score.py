"""Host side of the sampling / Inception-score harness: the statistics of
common/inception/inception_score.py:59-90 and the quantisation of SNGAN/gan_cifar_resnet.py:536-551.

The classifier itself (tfgan's frozen Inception graph, inception_score.py:29-47) needs downloaded weights, which this
environment does not have: `get_inception_score` takes it as a callable instead of building it."""
import numpy as np


def quantize_samples(samples, for_score=False):
    """Generator output in [-1, 1] -> int32 pixel values, as the reference does it on the host:
    `(x + 1) * (255 / 2)` for the sample grid (:538) and `(x + 1) * (255.99 / 2)` for the score (:551);
    `.astype('int32')` truncates toward zero."""
    s = np.asarray(samples, dtype=np.float32)
    return ((s + 1.) * ((255.99 if for_score else 255.) / 2)).astype('int32')


def preds2score(preds, splits):
    """exp(mean_x KL(p(y|x) || p(y))) per split of the class-probability rows, then (mean, std) over the splits
    (inception_score.py:59-66)."""
    preds = np.asarray(preds)
    n = preds.shape[0]
    scores = []
    for i in range(splits):
        part = preds[(i * n // splits):((i + 1) * n // splits), :]
        marginal = part.mean(axis=0, keepdims=True)
        kl = (part * (np.log(part) - np.log(marginal))).sum(axis=1).mean()
        scores.append(np.exp(kl))
    return np.mean(scores), np.std(scores)


def get_inception_score(images, splits=10, classifier=None, batch_size=64):
    """images: [N,H,W,3], either pixel values (max > 1.01: mapped to [-1,1] as :72-73) or already in [-1,1].
    `classifier(batch[-1,1] NHWC float32) -> logits [b, >=1000]` stands in for the Inception graph of :29-47;
    whole batches only, first 1000 logits, softmax, `preds2score` (:49-56, :86)."""
    images = np.asarray(images)
    if np.max(images[0]) > 1.01:
        images = 2 * (images / 255. - 0.5)
    assert images.ndim == 4 and images.shape[3] == 3, images.shape          # [batch, height, width, channel]
    assert np.max(images[0]) <= 1 and np.min(images[0]) >= -1
    if classifier is None:
        raise NotImplementedError('the Inception classifier needs downloaded weights (inception_score.py:29-47): pass '
                                  'classifier=callable returning logits')
    preds = []
    for i in range(len(images) // batch_size):
        logits = np.asarray(classifier(images[i * batch_size:(i + 1) * batch_size].astype(np.float32)))[:, :1000]
        e = np.exp(logits)
        preds.append(e / e.sum(axis=1, keepdims=True))
    return preds2score(np.concatenate(preds, 0), splits)
