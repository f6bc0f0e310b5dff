"""gan_lib_tensorflow_amd -- MI355X-native (gfx950) SNGAN-ResNet training hot path.

Mirror of the reference's import layout for this path:
    gan_lib_tensorflow_amd.common.ops.{conv2d,linear,sn,normalization,embedding,deconv2d}
    gan_lib_tensorflow_amd.common.resnet_block
    gan_lib_tensorflow_amd.SNGAN.gan_cifar_resnet   (Generator, Discriminator, SNGANTrainer)
All compute goes through libgank.so (include/gank.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
