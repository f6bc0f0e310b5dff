"""PGGAN, Nvidia architecture -- drop-in for PGGAN/model_nvidia.py of the reference (BASELINE.json config 4).

`PGGAN(args)` keeps the reference interface: `args.block_count` (number of up / down blocks: resolution 4 * 2**block_count),
`args.trans` (fade-in of the newest block), `args.inputs_norm`; `get_generator(z_var, alpha, training, reuse)`,
`get_discriminator(x_var, alpha, spectral_normed, update_collection, reuse)`, `generator_block`, `discriminator_block`,
`get_dim`, and the module functions `lrelu`, `minibatch_std` (model_nvidia.py:15-29), with the scopes (`g_net`, `d_net`) and
variable names of the reference.  Images are bf16 NHWC [N, H, W, 3].

Where the reference does not run as written, its intent is followed and said so here:
  * `lib.ops.pixelnorm.Pixelnorm` (:65,:70) does not exist -- `common.ops.normalization.pixel_norm` (:90,:96) is what the
    first two layers call and what every generator block means;
  * `get_dim` (:48) divides with `/` (a float under Python 3): the channel counts are its integer values;
  * `Conv2D(..., reuse=reuse)` (:147-160): Conv2D has no such argument -- variables are fetched by name, as everywhere.

MI355X-first: nearest-neighbour upsampling (`concat x4 + depth_to_space`, :58-59) never materialises -- the 3x3 conv that
follows it runs as the 4-phase transposed conv of the SNGAN path (4 instead of 9 taps per output); all spectral norms of a
critic pass are one batched launch group; pixel norm, leaky relu, fade-in blend, minibatch-std are single launches.
"""
from .. import functional as Fn
from .. import functional2 as F2
from ..common.ops import conv2d as _conv2d
from ..common.ops import linear as _linear
from ..common.ops import normalization as _norm
from ..common.ops import sn as _sn
from ..store import get_default_store


def lrelu(x, leakiness=0.2):
    assert leakiness <= 1, "leakiness must be <= 1"
    return Fn.relu(x, leakiness)                       # tf.maximum(x, leakiness * x)   (:15-17)


def minibatch_std(x):
    """(:20-29) x [B,H,W,C] -> [B,H,W,C+1]: one extra channel holding mean_{h,w,c} sqrt(var_batch(x) + 1e-8)"""
    return Fn.minibatch_std(x)


class PGGAN(object):
    def __init__(self, args):
        self.bc = args.block_count  # Count of up/down block.
        self.trans = args.trans  # If trans.
        self.inputs_norm = args.inputs_norm

    def get_dim(self, stage):
        return int(min(2048 // (2 ** stage), 512))

    def generator_block(self, inputs, out_dim, name='generator_block'):
        """(:50-73) NN-upsample, conv3x3 + pixel norm + lrelu, conv3x3 + pixel norm + lrelu"""
        store = get_default_store()
        with store.variable_scope(name):
            cin = inputs.shape[-1]
            output = _conv2d.Conv2D(inputs, cin, out_dim, 3, 1, 'Conv.1', inputs_norm=self.inputs_norm, he_init=True, biases=True,
                                    upsample=True)
            output = lrelu(_norm.pixel_norm(output))
            output = _conv2d.Conv2D(output, out_dim, out_dim, 3, 1, 'Conv.2', inputs_norm=self.inputs_norm, he_init=True, biases=True)
            output = lrelu(_norm.pixel_norm(output))
        return output

    def get_generator(self, z_var, alpha, training=True, reuse=False):
        """(:75-129) z_var [N, z_dim] bf16 -> images [N, 4 * 2**bc, 4 * 2**bc, 3]"""
        store = get_default_store()
        with store.variable_scope('g_net', reuse=reuse):
            z = z_var.reshape(z_var.shape[0], -1)
            output = _linear.Linear(z, z.shape[-1], 4 * 4 * 512, 'G.Input', inputs_norm=self.inputs_norm)
            output = output.reshape(-1, 4, 4, 512)
            output = lrelu(_norm.pixel_norm(output))
            output = _conv2d.Conv2D(output, 512, 512, 3, 1, 'G.Conv', inputs_norm=self.inputs_norm, he_init=True, biases=True)
            output = lrelu(_norm.pixel_norm(output))
            for i in range(self.bc - 1):
                output = self.generator_block(output, self.get_dim(i), 'G.UpBlock.{}'.format(i + 1))
            if self.trans:
                out_a, out_b = Fn.fork(output)
                toRGB1 = self.generator_block(out_a, self.get_dim(self.bc - 1), 'G.UpBlock.{}'.format(self.bc))
                toRGB1 = _conv2d.Conv2D(toRGB1, toRGB1.shape[-1], 3, 1, 1, 'G.{}_toRGB1'.format(self.bc),
                                        inputs_norm=self.inputs_norm, he_init=True, biases=True)
                # skip connection: the previous resolution's features, upsampled, through their own toRGB (:111-114)
                toRGB2 = _conv2d.Conv2D(out_b, out_b.shape[-1], 3, 1, 1, 'G.{}_toRGB2'.format(self.bc),
                                        inputs_norm=self.inputs_norm, he_init=True, biases=True, upsample=True)
                toRGB = Fn.blend(toRGB2, toRGB1, alpha)          # (1 - alpha) * toRGB2 + alpha * toRGB1   (:117)
            else:
                if self.bc > 0:
                    toRGB = self.generator_block(output, self.get_dim(self.bc - 1), 'G.UpBlock.{}'.format(self.bc))
                else:
                    toRGB = output
                toRGB = _conv2d.Conv2D(toRGB, toRGB.shape[-1], 3, 1, 1, 'G.{}_toRGB'.format(self.bc),
                                       inputs_norm=self.inputs_norm, he_init=True, biases=True)
        return toRGB

    def discriminator_block(self, inputs, out_dim, name, spectral_normed=False, update_collection=None, reuse=False):
        """(:131-162) conv3x3 + lrelu, conv3x3 + lrelu, 2x2 average pool"""
        store = get_default_store()
        with store.variable_scope(name):
            c = inputs.shape[-1]
            output = _conv2d.Conv2D(inputs, c, c, 3, 1, 'Conv.1', spectral_normed=spectral_normed,
                                    update_collection=update_collection, he_init=True, biases=True)
            output = lrelu(output)
            output = _conv2d.Conv2D(output, c, out_dim, 3, 1, 'Conv.2', spectral_normed=spectral_normed,
                                    update_collection=update_collection, he_init=True, biases=True)
            output = lrelu(output)
            output = Fn.meanpool2x2(output)
        return output

    def get_discriminator(self, x_var, alpha, spectral_normed=True, update_collection=None, reuse=False):
        """(:164-237) x_var [N, H, W, 3] -> logits [N]"""
        store = get_default_store()
        with store.variable_scope('d_net', reuse=reuse):
            prefix = store.full_name('')[:-1]
            ctx = _sn.precomputed(store, prefix, update_collection) if spectral_normed else _null()
            with ctx:
                kw = dict(spectral_normed=spectral_normed, update_collection=update_collection)
                if self.trans:
                    x_a, x_b = Fn.fork(x_var)
                    fromRGB1 = _conv2d.Conv2D(x_a, 3, self.get_dim(self.bc - 1), 1, 1, 'D.{}_fromRGB1'.format(self.bc),
                                              he_init=True, biases=True, **kw)
                    fromRGB1 = self.discriminator_block(fromRGB1, self.get_dim(self.bc - 1), 'D.Block.{}'.format(self.bc), **kw)
                    # skip connection (:196-203)
                    fromRGB2 = Fn.meanpool2x2(x_b)
                    fromRGB2 = _conv2d.Conv2D(fromRGB2, 3, self.get_dim(self.bc - 1), 1, 1, 'D.{}_fromRGB2'.format(self.bc),
                                              he_init=True, biases=True, **kw)
                    x_code = Fn.blend(fromRGB2, fromRGB1, alpha)        # (:207)
                else:
                    x_code = _conv2d.Conv2D(x_var, 3, self.get_dim(self.bc - 1), 1, 1, 'D.{}_fromRGB'.format(self.bc),
                                            he_init=True, biases=True, **kw)
                    if self.bc > 0:
                        x_code = self.discriminator_block(x_code, self.get_dim(self.bc - 1), 'D.Block.{}'.format(self.bc), **kw)
                for i in range(1, self.bc):
                    x_code = self.discriminator_block(x_code, self.get_dim(self.bc - 1 - i), 'D.Block.{}'.format(self.bc - i), **kw)
                output = minibatch_std(x_code)
                output = _conv2d.Conv2D(output, output.shape[-1], self.get_dim(self.bc - 1), 3, 1, 'D.Conv', he_init=True, biases=True, **kw)
                output = lrelu(output)
                output = F2.mean_hw(output)                               # tf.reduce_mean(output, axis=[1, 2])
                logits = _linear.Linear(output, output.shape[-1], 1, 'D.Output')
        return logits.reshape(-1)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
