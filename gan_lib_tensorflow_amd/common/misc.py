"""`get_loss` -- drop-in for common/misc.py:310-394 of the reference (the adversarial loss pair)."""
import torch

from .. import functional as Fn


def get_loss(disc_real, disc_fake, loss_type='HINGE'):
    """(d_loss, g_loss) for critic outputs on real / generated samples (misc.py:310-394).
      'HINGE'   (:326-335, the loss of the SNGAN and ACGAN scripts): mean(relu(1 - real)) + mean(relu(1 + fake)); -mean(fake)
      'WGAN'    (:328-336): -mean(real) + mean(fake); -mean(fake)
      'WGAN-GP' (:337-352): the same pair -- the reference leaves the penalty to the call site (ACGAN/train.py:99-107);
                use `gradient_penalty` below for it.
    One launch per loss computes the value and d loss / d logits.  The sigmoid-based branches (LSGAN, CGAN, MiniMax) are
    not used by any configuration of BASELINE.json."""
    n_real = disc_real.reshape(-1).shape[0]
    both = torch.cat([disc_real.reshape(-1), disc_fake.reshape(-1)], 0)
    if loss_type == 'HINGE':
        d_loss = Fn.hinge_d_loss(both, n_real)
    elif loss_type in ('WGAN', 'WGAN-GP'):
        d_loss = Fn.wgan_d_loss(both, n_real)
    else:
        raise NotImplementedError('loss_type %r: only HINGE / WGAN / WGAN-GP are on the configured paths' % (loss_type,))
    g_loss = Fn.hinge_g_loss(disc_fake.reshape(-1))        # -mean(disc_fake) in all three branches
    return d_loss, g_loss


def gradient_penalty(gradients, weight=10.0):
    """weight * mean((sqrt(sum(g^2, axis=[1,2,3]) + 1e-10) - 1)^2)  -- the block the reference asks to paste at the call
    site (misc.py:341-349; ACGAN/train.py:104-106).  `gradients` must come from a critic built of `functional2` operators
    with create_graph=True so the penalty can be differentiated with respect to the critic's weights."""
    from .. import functional2 as F2
    return F2.gradient_penalty(gradients, weight)
