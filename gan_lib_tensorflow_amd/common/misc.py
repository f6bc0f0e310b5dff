"""`get_loss` -- drop-in for common/misc.py:310-394 of the reference (the adversarial loss pair)."""
import torch

from .. import functional as Fn


def get_loss(disc_real, disc_fake, loss_type='HINGE'):
    """(d_loss, g_loss) for critic outputs on real / generated samples.  'HINGE' (misc.py:326-335, the loss of the
    SNGAN and ACGAN scripts) runs on the fused loss+gradient kernels; the other branches of the reference
    (WGAN, WGAN-GP, LSGAN, CGAN, MiniMax) are not on the hot path."""
    if loss_type != 'HINGE':
        raise NotImplementedError('only the HINGE branch (misc.py:326-335) is on the hot path; got %r' % (loss_type,))
    n_real = disc_real.reshape(-1).shape[0]
    both = torch.cat([disc_real.reshape(-1), disc_fake.reshape(-1)], 0)
    # mean(relu(1 - real)) + mean(relu(1 + fake)): one launch computes the value and d loss / d logits
    d_loss = Fn.hinge_d_loss(both, n_real)
    g_loss = Fn.hinge_g_loss(disc_fake.reshape(-1))
    return d_loss, g_loss
