"""ACGAN train step -- the loop body of ACGAN/train.py:89-201 of the reference (BASELINE.json config 3).

One step (train.py:194-204) = 1 generator update (skipped at step 0) + n_dis = 5 critic updates:
    critic loss    = hinge(D(real), D(G(z, fake_labels)))                                   (misc.get_loss, :96)
                   + 10 * mean((||grad_x D(x_hat)||_2 - 1)^2),  x_hat = real + alpha (fake - real)   (:99-107)
                   + mean softmax cross-entropy of the class head on the real batch          (:111-114)
    generator loss = -mean(D(G(z))) + acgan_scale_G * cross-entropy of the class head on the fakes (:117-121)
both with tf.train.AdamOptimizer(beta1=0, beta2=0.9) at a learning rate that decays linearly from 4e-4 to 2e-4 over
max_iter / 2 generator steps (tf.train.polynomial_decay on `global_step`, :141-148).

Data parallel (config 3: global batch 256 over 8 GPUs): one process per GPU, each rank a replica with batch/world
samples and its own batch-norm statistics (as the reference's towers would have); flat gradient buffers summed with one
RCCL all-reduce per update, 1/world applied inside the Adam kernel.
"""
import numpy as np
import torch

from .. import functional as Fn
from .. import functional2 as F2
from .. import kernels as K
from .. import parallel
from ..graphs import GraphRunner
from ..store import ParamStore, set_default_store
from .model import ACGAN


def polynomial_decay(step, lr0=0.0004, decay_steps=50000, lr_end=0.0002):
    """tf.train.polynomial_decay(power=1, cycle=False)   (train.py:141)"""
    s = min(step, decay_steps)
    return (lr0 - lr_end) * (1.0 - s / float(decay_steps)) + lr_end


BATCHED_PREP = True     # operand copies of the weights: one launch per network and update (False: per conv and layout on first use)


def _g_prep_kind(name, W):
    """operand layout per generator weight (kernels.prep_weights_batched), by the kernel its conv wrapper picks: the UpsampleConv
    3x3 layers run phase-decomposed, the 16x16 conv of G.2 on the image-resident kernel where its shape allows; a wrapper that
    wants another layout than the one attached prepares its own."""
    if W.dim() != 4:
        return None
    # (the same choices as the SNGAN generator's, whose blocks these are: SNGAN/gan_cifar_resnet.py _g_prep_kind)
    if Fn.RES8_CONV and name.endswith(('G.1.Conv1/Filters', 'G.1.Conv2/Filters')) and W.shape[0] == 3 and W.shape[2] in (128, 256) and W.shape[3] % 128 == 0:
        return 4                      # 8x8 images: LDS-resident kernel, fragment-major operands
    if Fn.PHASE_UPCONV and name.endswith('.Conv1/Filters') and W.shape[0] == 3:
        return 1
    if Fn.IMG16_CONV and name.endswith('G.2.Conv2/Filters') and W.shape[0] == 3 and W.shape[2] % 128 == 0 and W.shape[3] % 128 == 0:
        return 4                      # 16x16 image-resident conv
    return 0


class ACGANTrainer:
    def __init__(self, batch_size=64, z_dim=128, acgan_scale_G=0.1, n_dis=5, max_iter=100000, device="cuda", seed=0,
                 process_group=None, state=None, use_graphs=True, allow_eager_fallback=False):
        self.device = torch.device(device)
        self.batch, self.z_dim, self.scale_g, self.n_dis, self.max_iter = batch_size, z_dim, acgan_scale_G, n_dis, max_iter
        self.store = set_default_store(ParamStore(self.device, seed=seed))
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.rng_state = K.new_rng_state(parallel.data_seed(seed, self.rank), self.device)
        self.model = ACGAN()
        self.global_step = 0
        # build once (variables are created by name on first use)
        with torch.no_grad():
            labels = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
            z = torch.zeros((batch_size, z_dim), dtype=K.BF16, device=self.device)
            x = self.model.get_generator(z, labels)
            self.model.get_discriminator(x, labels)
        if state is not None:
            self.store.load_state_dict(state)
        self.g_flat = self.store.flatten('g_net')
        self.d_flat = self.store.flatten('d_net')
        self.g_params = [self.store.vars[k] for k in self.g_flat['names']]
        self.d_params = [self.store.vars[k] for k in self.d_flat['names']]
        for p in self.d_params:
            p.grad = p.main_grad         # the twice-differentiable critic RETURNS weight gradients: autograd adds them in place here
            del p.main_grad
        self.g_opt = self._adam(self.g_flat)
        self.d_opt = self._adam(self.d_flat)
        self._g_convs = [(k, v) for k, v in self.store.vars.items() if k.startswith('g_net/') and k.endswith('/Filters') and v.dim() == 4]
        self._refresh_g_prep()
        self.losses = {}
        # the two updates as captured hipGraphs (gan_lib_tensorflow_amd/graphs.py): static input buffers, the learning rate
        # written into the optimiser's device-side hyper-parameters outside the captured region
        self.graphs = GraphRunner(use_graphs, allow_eager_fallback)     # a failed hipGraph capture raises unless the caller allows eager execution
        self.real_u8 = torch.zeros((batch_size, 3072), dtype=torch.uint8, device=self.device)
        self.real_labels = torch.zeros(batch_size, dtype=torch.int32, device=self.device)

    def _adam(self, flat):
        dev = self.device
        return dict(hp=torch.tensor([0.0004, 0.0, 0.9, 1e-8, 1.0 / self.world, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev),
                    t=torch.zeros(1, dtype=torch.int64, device=dev), flat=flat)

    def _set_lr(self, opt):
        opt['hp'][0:1].fill_(polynomial_decay(self.global_step, decay_steps=self.max_iter // 2))

    def _apply(self, opt):
        f = opt['flat']
        K.adam_tf(f['params'], f['grads'], f['m'], f['v'], opt['hp'], opt['t'], None)
        if opt is self.g_opt:
            self._refresh_g_prep()

    def _refresh_g_prep(self):
        """The generator's MFMA operand copies, once per generator update in ONE launch (every pass until the next update -- the
        fakes of the critic updates, the generator's own forward / backward -- picks them up from the variables) instead of one
        or two launches per conv and pass."""
        if BATCHED_PREP and self._g_convs:
            K.prep_weights_batched([v for _, v in self._g_convs], want_d=True, kinds=[_g_prep_kind(k, v) for k, v in self._g_convs])

    def _update(self, key, fwd_bwd, opt):
        """fwd_bwd (graph) -> [RCCL all-reduce] -> Adam (graph): one graph when there is nothing to exchange"""
        self._set_lr(opt)
        if self.world == 1:
            self.graphs.run(key, lambda: (fwd_bwd(), self._apply(opt)))
        else:
            self.graphs.run(key, fwd_bwd)
            parallel.allreduce_sum_(opt['flat']['grads'], self.pg)
            self.graphs.run(key + '/adam', lambda: self._apply(opt))

    # ---- the two losses (eager; explicit inputs override the device RNG for parity tests) -----------------------------
    def d_loss(self, real, real_labels, z=None, fake_labels=None, alpha=None):
        set_default_store(self.store)
        b = real.shape[0]
        m = self.model
        if z is None:
            z = K.rng_normal((b, self.z_dim), self.rng_state)
        if fake_labels is None:
            fake_labels = K.rng_labels(b, 10, self.rng_state)
        if alpha is None:
            alpha = K.rng_uniform(b, self.rng_state)
        with torch.no_grad():            # d_train_op differentiates w.r.t. d_vars only (train.py:148)
            x_fake = m.get_generator(z, fake_labels)
        disc_real, ac_real = m.get_discriminator(real, real_labels, update_collection=None)
        disc_fake, _ = m.get_discriminator(x_fake, fake_labels, update_collection='NO_OPS', reuse=True)
        d_gan = Fn.hinge_d_loss(Fn.concat_rows(disc_real, disc_fake), b)
        interp = K.lerp_rows(real, x_fake, alpha).requires_grad_(True)
        d_int, _ = m.get_discriminator(interp, real_labels, 'NO_OPS', reuse=True)
        ones = Fn.constant_like(d_int, 1.0)                 # tf.gradients(D(x_hat), [x_hat]): d(sum of logits)/d(x_hat); a persistent buffer
        with F2.input_gradient_only():       # the filter / bias / table gradients of this pass are not part of the penalty
            (grads,) = torch.autograd.grad([d_int], [interp], [ones], create_graph=True)
        gp = F2.gradient_penalty(grads, 10.0)
        d_ac = Fn.softmax_xent(ac_real, real_labels)
        self.losses.update(d_loss_gan=K.weighted_sum_f32([d_gan.detach(), gp.detach()], [1.0, 1.0]), d_loss_acgan=d_ac.detach(), gradient_penalty=gp.detach())
        return Fn.weighted_sum([d_gan, gp, d_ac])

    def g_loss(self, z=None, fake_labels=None):
        set_default_store(self.store)
        b = self.batch
        m = self.model
        if z is None:
            z = K.rng_normal((b, self.z_dim), self.rng_state)
        if fake_labels is None:
            fake_labels = K.rng_labels(b, 10, self.rng_state)
        x_fake = m.get_generator(z, fake_labels)
        for p in self.d_params:          # g_train_op differentiates w.r.t. g_vars only (train.py:146)
            p.requires_grad_(False)
        try:
            disc_fake, ac_fake = m.get_discriminator(x_fake, fake_labels, update_collection='NO_OPS', reuse=True)
            g_gan = Fn.hinge_g_loss(disc_fake)
            g_ac = Fn.softmax_xent(ac_fake, fake_labels)
            total = Fn.weighted_sum([g_gan, g_ac], [1.0, self.scale_g])
        finally:
            for p in self.d_params:
                p.requires_grad_(True)
        self.losses.update(g_loss_gan=g_gan.detach(), g_loss_acgan=g_ac.detach())
        return total

    # ---- updates --------------------------------------------------------------------------------------------------
    def _d_fwd_bwd(self):
        with F2.one_update():            # every pass over the critic in this update shares one preparation of its weights
            if BATCHED_PREP:
                F2.prepare_batched(self.d_params)
            real = K.preprocess_real(self.real_u8, self.rng_state)         # [B, 32, 32, 3] bf16   (train.py:80-83)
            K.zero_(self.d_flat['grads'])
            loss = self.d_loss(real, self.real_labels)
            loss.backward(gradient=Fn.unit_seed(loss))
            self.losses['d_loss'] = loss.detach()

    def _g_fwd_bwd(self):
        with F2.one_update():
            if BATCHED_PREP:
                F2.prepare_batched(self.d_params)
            self.store.zero_grads('g_net')
            loss = self.g_loss()
            loss.backward(gradient=Fn.unit_seed(loss))
            self.losses['g_loss'] = loss.detach()

    def d_step(self, real_u8, labels):
        """one critic update on a uint8 [B, 3072] CHW-planar batch + int labels (train.py:199-204)"""
        self.real_u8.copy_(real_u8, non_blocking=True)
        self.real_labels.copy_(labels, non_blocking=True)
        self._update('d', self._d_fwd_bwd, self.d_opt)
        return self.losses['d_loss']

    def g_step(self):
        self._update('g', self._g_fwd_bwd, self.g_opt)
        self.global_step += 1            # minimize(..., global_step=global_step) on the generator's optimiser (train.py:146)
        return self.losses['g_loss']

    def train_iteration(self, batches, step=None):
        step = self.global_step_counter if step is None else step
        if step > 0:
            self.g_step()
        for _ in range(self.n_dis):
            data, labels = next(batches)
            self.d_step(data, labels)
        self.global_step_counter = step + 1

    global_step_counter = 0
