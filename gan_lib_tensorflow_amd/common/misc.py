"""`get_loss` -- drop-in for common/misc.py:310-394 of the reference (the adversarial loss pair)."""
import torch

from .. import functional as Fn


def get_loss(disc_real, disc_fake, loss_type='HINGE'):
    """(d_loss, g_loss) for critic outputs on real / generated samples (misc.py:310-394).
      'HINGE'   (:326-335, the loss of the SNGAN and ACGAN scripts): mean(relu(1 - real)) + mean(relu(1 + fake)); -mean(fake)
      'WGAN'    (:328-336): -mean(real) + mean(fake); -mean(fake)
      'WGAN-GP' (:337-352): the same pair -- the reference leaves the penalty to the call site (ACGAN/train.py:99-107);
                use `gradient_penalty` below for it.
      'LSGAN'   (:353-360): (mean((1 - real)^2) + mean(fake^2)) / 2; mean((1 - fake)^2) / 2
      'CGAN'    (:361-372): sigmoid cross-entropy against ones / zeros; generator: against ones on the fakes
      'Modified_MiniMax' (:373-381): -mean(log sigmoid(real)) - mean(log(1 - sigmoid(fake))); -mean(log sigmoid(fake))
      'MiniMax' (:382-390): the same critic loss; generator mean(log(1 - sigmoid(fake)))
    One launch per loss computes the value and d loss / d logits (the sigmoid family through the stable softplus form)."""
    n_real = disc_real.reshape(-1).shape[0]
    fake = disc_fake.reshape(-1)
    both = Fn.concat_rows(disc_real.reshape(-1), fake)
    if loss_type == 'HINGE':
        return Fn.hinge_d_loss(both, n_real), Fn.hinge_g_loss(fake)
    if loss_type in ('WGAN', 'WGAN-GP'):
        return Fn.wgan_d_loss(both, n_real), Fn.hinge_g_loss(fake)        # -mean(disc_fake) in all three branches
    if loss_type == 'LSGAN':
        return Fn.gan_pointwise_loss(both, n_real, 0), Fn.gan_pointwise_loss(fake, 0, 1)
    if loss_type in ('CGAN', 'Modified_MiniMax'):
        return Fn.gan_pointwise_loss(both, n_real, 2), Fn.gan_pointwise_loss(fake, 0, 3)
    if loss_type == 'MiniMax':
        return Fn.gan_pointwise_loss(both, n_real, 2), Fn.gan_pointwise_loss(fake, 0, 4)
    raise NotImplementedError('loss_type %r (misc.py:310-394 knows HINGE, WGAN, WGAN-GP, LSGAN, CGAN, Modified_MiniMax, MiniMax)' % (loss_type,))


def gradient_penalty(gradients, weight=10.0):
    """weight * mean((sqrt(sum(g^2, axis=[1,2,3]) + 1e-10) - 1)^2)  -- the block the reference asks to paste at the call
    site (misc.py:341-349; ACGAN/train.py:104-106).  `gradients` must come from a critic built of `functional2` operators
    with create_graph=True so the penalty can be differentiated with respect to the critic's weights."""
    from .. import functional2 as F2
    return F2.gradient_penalty(gradients, weight)
