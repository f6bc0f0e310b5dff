from . import conv2d, deconv2d, embedding, linear, normalization, sn  # noqa: F401
