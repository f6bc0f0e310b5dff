"""Pix2Pix model -- drop-in for Pix2Pix/model.py of the reference: `Pix2Pix().get_generator(...)` / `.get_discriminator(...)`
with the reference's arguments and scopes (`g_net`, `d_net`); net_type 'UNet' (BASELINE.json config 5) is built, the other
branches of the reference (attention U-Net, ResNet, VGG19 with downloaded weights) raise as unknown types do there."""
from ..common.ops import sn as _sn
from ..store import get_default_store
from . import networks


class Pix2Pix(object):
    def __init__(self):
        pass

    def get_generator(self, inputs, outputs_channels, ngf=64, conv_type='conv2d', channel_multiplier=None,
                      padding='SAME', net_type='UNet', reuse=False, upsampe_method='depth_to_space', rng_state=None):
        """model.py:15-59.  rng_state: the device RNG the decoder's dropout draws from"""
        store = get_default_store()
        with store.variable_scope('g_net', reuse=reuse):
            if net_type == 'UNet':
                return networks.unet_generator(inputs, outputs_channels, ngf, conv_type=conv_type, channel_multiplier=channel_multiplier,
                                               padding=padding, upsampe_method=upsampe_method, rng_state=rng_state)
            raise NotImplementedError('Generator model name [%s] is not recognized' % net_type)

    def get_discriminator(self, inputs, targets, ndf=64, spectral_normed=True, update_collection=None,
                          conv_type='conv2d', channel_multiplier=None, padding='VALID', net_type='UNet', reuse=False):
        """model.py:61-103: inputs = real A image, targets = real B image or the generator's output"""
        store = get_default_store()
        with store.variable_scope('d_net', reuse=reuse):
            if net_type != 'UNet':
                raise NotImplementedError('Discriminator model name [%s] is not recognized' % net_type)
            prefix = store.full_name('')[:-1]
            # all six spectral norms of a critic pass: one batched launch group
            if spectral_normed:
                with _sn.precomputed(store, prefix, update_collection):
                    return networks.unet_discriminator(inputs, targets, ndf, spectral_normed, update_collection, conv_type=conv_type,
                                                       channel_multiplier=channel_multiplier, padding=padding)
            return networks.unet_discriminator(inputs, targets, ndf, spectral_normed, update_collection, conv_type=conv_type,
                                               channel_multiplier=channel_multiplier, padding=padding)
