"""Pix2Pix train step -- create_model and the loop body of Pix2Pix/train.py:447-568,700-730 of the reference (config 5).

    outputs       = G(inputs)                                             U-Net, SAME padding, dropout in the decoder
    discrim_loss  = mean(relu(1 - D(inputs, targets))) + mean(relu(1 + D(inputs, outputs)))     (misc.get_loss 'HINGE')
    gen_loss      = gan_weight * (-mean(D(inputs, outputs))) + l1_weight * mean(|targets - outputs|)
both critic passes run with update_collection=None (the spectral-norm `u` advances in each, train.py:458-475); one step =
n_dis critic updates then one generator update (:704-730); tf.train.AdamOptimizer(beta1=0, beta2=0.9) at a learning rate
decaying linearly from initial_lr to end_lr over max_steps generator steps (polynomial_decay on global_step, :521-541).
`args`: any object with the reference's flag names (batch_size, ngf, ndf, l1_weight, gan_weight, initial_lr, end_lr,
max_steps, n_dis, conv_type, upsampe_method).
"""
import types

import torch

from .. import functional as Fn
from .. import kernels as K
from .. import parallel
from ..graphs import GraphRunner
from ..store import ParamStore, set_default_store
from .model import Pix2Pix


def default_args(**over):
    """the argparse defaults of train.py:29-80"""
    a = dict(batch_size=64, ngf=64, ndf=64, l1_weight=100.0, gan_weight=1.0, initial_lr=0.0002, end_lr=0.0001, beta1=0., beta2=0.9,
             max_steps=100000, n_dis=5, conv_type='conv2d', channel_multiplier=0, net_type='UNet', upsampe_method='depth_to_space',
             loss_type='HINGE', crop_size=256)
    a.update(over)
    return types.SimpleNamespace(**a)


def polynomial_decay(step, lr0, decay_steps, lr_end):
    s = min(step, decay_steps)
    return (lr0 - lr_end) * (1.0 - s / float(decay_steps)) + lr_end


class Pix2PixTrainer:
    def __init__(self, args, device="cuda", seed=0, process_group=None, state=None, in_channels=3, out_channels=3, use_graphs=True, allow_eager_fallback=False):
        if args.loss_type != 'HINGE':
            raise NotImplementedError('loss_type HINGE (the reference default, train.py:38)')
        self.args = args
        self.device = torch.device(device)
        self.store = set_default_store(ParamStore(self.device, seed=seed))
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.rng_state = K.new_rng_state(parallel.data_seed(seed, self.rank), self.device)
        self.model = Pix2Pix()
        self.out_channels = out_channels
        self.global_step = 0
        with torch.no_grad():            # build once: variables are created by name on first use
            s = args.crop_size
            a = torch.zeros((args.batch_size, s, s, in_channels), dtype=K.BF16, device=self.device)
            out = self._generator(a, reuse=False)
            self._critic(a, out, 'NO_OPS', reuse=False)
        if state is not None:
            self.store.load_state_dict(state)
        self.g_flat = self.store.flatten('g_net')
        self.d_flat = self.store.flatten('d_net')
        self.g_params = [self.store.vars[k] for k in self.g_flat['names']]
        self.d_params = [self.store.vars[k] for k in self.d_flat['names']]
        self.g_opt = self._adam(self.g_flat)
        self.d_opt = self._adam(self.d_flat)
        self.losses = {}
        # the two updates as captured hipGraphs: static input / target buffers, the learning rate written outside the capture
        self.graphs = GraphRunner(use_graphs, allow_eager_fallback)     # a failed hipGraph capture raises unless the caller allows eager execution
        self.inputs = torch.zeros((args.batch_size, args.crop_size, args.crop_size, in_channels), dtype=K.BF16, device=self.device)
        self.targets = torch.zeros((args.batch_size, args.crop_size, args.crop_size, out_channels), dtype=K.BF16, device=self.device)

    def _generator(self, inputs, reuse=True):
        a = self.args
        return self.model.get_generator(inputs, self.out_channels, ngf=a.ngf, conv_type=a.conv_type, channel_multiplier=a.channel_multiplier,
                                        padding='SAME', net_type=a.net_type, reuse=reuse, upsampe_method=a.upsampe_method, rng_state=self.rng_state)

    def _critic(self, inputs, targets, update_collection, reuse=True):
        a = self.args
        return self.model.get_discriminator(inputs, targets, ndf=a.ndf, spectral_normed=True, update_collection=update_collection,
                                            conv_type=a.conv_type, channel_multiplier=a.channel_multiplier, padding='VALID',
                                            net_type=a.net_type, reuse=reuse)

    def _adam(self, flat):
        dev = self.device
        return dict(hp=torch.tensor([self.args.initial_lr, self.args.beta1, self.args.beta2, 1e-8, 1.0 / self.world, 0.0, 0.0, 0.0],
                                    dtype=torch.float32, device=dev),
                    t=torch.zeros(1, dtype=torch.int64, device=dev), flat=flat)

    def _apply(self, opt):
        f = opt['flat']
        K.adam_tf(f['params'], f['grads'], f['m'], f['v'], opt['hp'], opt['t'], None, zero_grads=True)

    def _update(self, key, fwd_bwd, opt):
        """fwd_bwd (graph) -> [RCCL all-reduce] -> Adam (graph): one graph when there is nothing to exchange"""
        a = self.args
        opt['hp'][0:1].fill_(polynomial_decay(self.global_step, a.initial_lr, a.max_steps, a.end_lr))
        if self.world == 1:
            self.graphs.run(key, lambda: (fwd_bwd(), self._apply(opt)))
        else:
            self.graphs.run(key, fwd_bwd)
            parallel.allreduce_sum_(opt['flat']['grads'], self.pg)
            self.graphs.run(key + '/adam', lambda: self._apply(opt))

    # ---- losses ---------------------------------------------------------------------------------------------------
    def d_loss(self, inputs, targets):
        """discrim_loss (train.py:477-483); the generator is not differentiated (var_list=discrim_tvars, :546)"""
        set_default_store(self.store)
        with torch.no_grad():
            outputs = self._generator(inputs)
        predict_real = self._critic(inputs, targets, None)
        predict_fake = self._critic(inputs, outputs, None)
        n = predict_real.numel()
        return Fn.hinge_d_loss(Fn.concat_rows(predict_real.reshape(-1), predict_fake.reshape(-1)), n)

    def g_loss(self, inputs, targets):
        """gen_loss (train.py:504-512); the critic's variables are not differentiated (var_list=gen_tvars, :552).  The
        generator update's graph contains BOTH critic passes with update_collection=None (:452-475), so `u` advances twice."""
        set_default_store(self.store)
        outputs, outputs_l1 = Fn.fork(self._generator(inputs))     # read by the critic and by the L1 term: one add launch backward
        for p in self.d_params:
            p.requires_grad_(False)
        try:
            with torch.no_grad():
                self._critic(inputs, targets, None)            # predict_real: only its u update is observable here
            predict_fake = self._critic(inputs, outputs, None)
            gan = Fn.hinge_g_loss(predict_fake.reshape(-1))
            l1 = Fn.l1_loss(outputs_l1, targets)
        finally:
            for p in self.d_params:
                p.requires_grad_(True)
        self.losses.update(gen_loss_GAN=gan.detach(), gen_loss_L1=l1.detach())
        return Fn.weighted_sum([gan, l1], [self.args.gan_weight, self.args.l1_weight])

    # ---- updates --------------------------------------------------------------------------------------------------
    def _backward(self, loss):
        Fn.reset_deferred()
        try:
            loss.backward(gradient=Fn.unit_seed(loss))
            Fn.join_wgrad()
        finally:
            Fn.reset_deferred()

    def _d_fwd_bwd(self):
        loss = self.d_loss(self.inputs, self.targets)
        self._backward(loss)
        self.losses['discrim_loss'] = loss.detach()

    def _g_fwd_bwd(self):
        loss = self.g_loss(self.inputs, self.targets)
        self._backward(loss)
        self.losses['gen_loss'] = loss.detach()

    def _feed(self, inputs, targets):
        if inputs.data_ptr() != self.inputs.data_ptr():
            self.inputs.copy_(inputs, non_blocking=True)
        if targets.data_ptr() != self.targets.data_ptr():
            self.targets.copy_(targets, non_blocking=True)

    def d_step(self, inputs, targets):
        self._feed(inputs, targets)
        self._update('d', self._d_fwd_bwd, self.d_opt)
        return self.losses['discrim_loss']

    def g_step(self, inputs, targets):
        self._feed(inputs, targets)
        self._update('g', self._g_fwd_bwd, self.g_opt)
        self.global_step += 1            # apply_gradients(..., global_step=global_step) on the generator's optimiser (:554)
        return self.losses['gen_loss']

    def train_step(self, inputs, targets):
        """train.py:704-730: n_dis critic updates, then the generator update, on one batch of pairs"""
        for _ in range(self.args.n_dis):
            self.d_step(inputs, targets)
        return self.g_step(inputs, targets)
