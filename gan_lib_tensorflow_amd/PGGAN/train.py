"""PGGAN train step -- the loop body of PGGAN/train.py:61-219 of the reference (BASELINE.json config 4).

One step (train.py:185-193) = 1 generator update + n_dis = 5 critic updates at fade-in weight alpha = step / max_iter:
    real images : CIFAR-10 rows -> 2 (x/256 - .5) + U[0, 1/128) -> NHWC -> bilinear resize to image_size (through
                  image_size/2 first while a new block fades in, :88-92)
    critic loss : mean(relu(1 - D(real))) + mean(relu(1 + D(G(z))))       (:104-106)   D(real) updates the spectral-norm u
    gen loss    : -mean(D(G(z)))                                           (:107)       (D(fake) runs with NO_OPS, :102)
both with tf.train.AdamOptimizer(1e-4, beta1=0, beta2=0.9)               (:130-134).
`args` is any object with the reference's flag names (batch_size, image_size, block_count, trans, inputs_norm, z_dim, n_dis,
max_iter).  Data parallel: flat gradient buffers, one RCCL all-reduce per update, 1/world inside the Adam kernel.
"""
import types

import torch

from .. import functional as Fn
from .. import kernels as K
from .. import parallel
from ..graphs import GraphRunner
from ..store import ParamStore, set_default_store
from .model_nvidia import PGGAN


def default_args(**over):
    """the argparse defaults of train.py:24-57"""
    a = dict(batch_size=16, image_size=4, max_iter=100000, n_dis=5, z_dim=512, image_dim=3072, model='nvidia', block_count=0,
             trans=False, inputs_norm=False)
    a.update(over)
    return types.SimpleNamespace(**a)


class PGGANTrainer:
    def __init__(self, args, device="cuda", seed=0, process_group=None, state=None, use_graphs=True, allow_eager_fallback=False):
        assert args.image_size == 4 * 2 ** args.block_count, "image_size must be 4 * 2**block_count (train.py:52-54)"
        self.args = args
        self.device = torch.device(device)
        self.store = set_default_store(ParamStore(self.device, seed=seed))
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.rng_state = K.new_rng_state(parallel.data_seed(seed, self.rank), self.device)
        self.model = PGGAN(args)
        self.step = 0
        with torch.no_grad():            # build once: variables are created by name on first use
            z = torch.zeros((args.batch_size, args.z_dim), dtype=K.BF16, device=self.device)
            x = self.model.get_generator(z, 0.0)
            self.model.get_discriminator(x, 0.0, update_collection='NO_OPS')
        if state is not None:
            self.store.load_state_dict(state)
        self.g_flat = self.store.flatten('g_net')
        self.d_flat = self.store.flatten('d_net')
        self.g_params = [self.store.vars[k] for k in self.g_flat['names']]
        self.d_params = [self.store.vars[k] for k in self.d_flat['names']]
        self.g_opt = self._adam(self.g_flat)
        self.d_opt = self._adam(self.d_flat)
        self.losses = {}
        # the two updates as captured hipGraphs: static input rows, the fade-in weight in device memory (written before a replay)
        self.graphs = GraphRunner(use_graphs, allow_eager_fallback)     # a failed hipGraph capture raises unless the caller allows eager execution
        self.real_u8 = torch.zeros((args.batch_size, args.image_dim), dtype=torch.uint8, device=self.device)
        self.alpha_dev = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _adam(self, flat):
        dev = self.device
        return dict(hp=torch.tensor([0.0001, 0.0, 0.9, 1e-8, 1.0 / self.world, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev),
                    t=torch.zeros(1, dtype=torch.int64, device=dev), flat=flat)

    def _apply(self, opt):
        f = opt['flat']
        K.adam_tf(f['params'], f['grads'], f['m'], f['v'], opt['hp'], opt['t'], None, zero_grads=True)

    def _update(self, key, fwd_bwd, opt):
        """fwd_bwd (graph) -> [RCCL all-reduce] -> Adam (graph): one graph when there is nothing to exchange"""
        if self.world == 1:
            self.graphs.run(key, lambda: (fwd_bwd(), self._apply(opt)))
        else:
            self.graphs.run(key, fwd_bwd)
            parallel.allreduce_sum_(opt['flat']['grads'], self.pg)
            self.graphs.run(key + '/adam', lambda: self._apply(opt))

    def alpha(self, step=None):
        return float(self.step if step is None else step) / float(self.args.max_iter)        # feed_dict alpha (:186)

    # ---- inputs ---------------------------------------------------------------------------------------------------
    def real_images(self, real_u8):
        """uint8 [B, 3072] CHW-planar rows -> bf16 [B, image_size, image_size, 3]   (:81-92)"""
        a = self.args
        x = K.preprocess_real(real_u8, self.rng_state)                          # [B, 32, 32, 3]
        if a.trans and a.block_count:
            x = K.resize_bilinear(x, (a.image_size // 2, a.image_size // 2))
        if x.shape[1] != a.image_size:
            x = K.resize_bilinear(x, (a.image_size, a.image_size))
        return x

    # ---- the two losses (explicit inputs override the device RNG for parity tests) -------------------------------
    def d_loss(self, real, z=None, alpha=None):
        set_default_store(self.store)
        b = real.shape[0]
        alpha = self.alpha() if alpha is None else alpha
        if z is None:
            z = K.rng_normal((b, self.args.z_dim), self.rng_state)
        with torch.no_grad():            # d_train_op differentiates w.r.t. d_vars only (:134)
            x_fake = self.model.get_generator(z, alpha, reuse=True)
        disc_real = self.model.get_discriminator(real, alpha, update_collection=None, reuse=True)
        disc_fake = self.model.get_discriminator(x_fake, alpha, update_collection='NO_OPS', reuse=True)
        return Fn.hinge_d_loss(Fn.concat_rows(disc_real, disc_fake), b)

    def g_loss(self, z=None, alpha=None):
        set_default_store(self.store)
        alpha = self.alpha() if alpha is None else alpha
        if z is None:
            z = K.rng_normal((self.args.batch_size, self.args.z_dim), self.rng_state)
        x_fake = self.model.get_generator(z, alpha, reuse=True)
        for p in self.d_params:          # g_train_op differentiates w.r.t. g_vars only (:132)
            p.requires_grad_(False)
        try:
            disc_fake = self.model.get_discriminator(x_fake, alpha, update_collection='NO_OPS', reuse=True)
            loss = Fn.hinge_g_loss(disc_fake)
        finally:
            for p in self.d_params:
                p.requires_grad_(True)
        return loss

    # ---- updates --------------------------------------------------------------------------------------------------
    def _backward(self, loss):
        Fn.reset_deferred()
        try:
            loss.backward(gradient=Fn.unit_seed(loss))
            Fn.join_wgrad()
        finally:
            Fn.reset_deferred()

    def _d_fwd_bwd(self):
        loss = self.d_loss(self.real_images(self.real_u8), alpha=self.alpha_dev)
        self._backward(loss)
        self.losses['d_loss'] = loss.detach()

    def _g_fwd_bwd(self):
        loss = self.g_loss(alpha=self.alpha_dev)
        self._backward(loss)
        self.losses['g_loss'] = loss.detach()

    def d_step(self, real_u8, alpha=None):
        self.real_u8.copy_(real_u8, non_blocking=True)
        self.alpha_dev.fill_(self.alpha() if alpha is None else float(alpha))
        self._update('d', self._d_fwd_bwd, self.d_opt)
        return self.losses['d_loss']

    def g_step(self, alpha=None):
        self.alpha_dev.fill_(self.alpha() if alpha is None else float(alpha))
        self._update('g', self._g_fwd_bwd, self.g_opt)
        return self.losses['g_loss']

    def train_iteration(self, batches):
        """one pass of the loop at train.py:185-193"""
        a = self.alpha()
        self.g_step(a)
        for _ in range(self.args.n_dis):
            data, _labels = next(batches)
            self.d_step(data, a)
        self.step += 1

    @torch.no_grad()
    def sample(self, n=100, z=None, alpha=None):
        """fixed-noise samples (:138-145)"""
        set_default_store(self.store)
        if z is None:
            z = K.rng_normal((n, self.args.z_dim), self.rng_state)
        return self.model.get_generator(z, self.alpha() if alpha is None else alpha, reuse=True)
