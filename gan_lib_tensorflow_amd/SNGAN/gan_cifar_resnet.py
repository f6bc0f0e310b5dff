"""SNGAN-ResNet CIFAR-10 -- drop-in for the model and train-step part of SNGAN/gan_cifar_resnet.py.

`Generator(n_samples, labels, noise=None, reuse=False)` and
`Discriminator(inputs, labels, update_collection=None, reuse=False)` keep the reference signatures
(:237, :266) and variable names; `SNGANTrainer` is the loop body (:599-620) with the graph the script
builds at import time (:317-526): 1 generator update on 2x64 fakes + N_CRITIC=5 critic updates on
64 real + 64 fake, hinge loss, TF-Adam(beta1=0, beta2=0.9), LR decay.

MI355X-first structure
  * the reference's "two towers on one GPU" hack (:73-75) becomes a `groups` argument: one launch over
    the whole batch, conditional-batch-norm statistics per tower -- identical arithmetic;
  * the 12 spectral norms of the critic run as one batched launch group per critic forward;
  * all trainable variables of a network live in one flat fp32 buffer: one memset zeroes the
    gradients, one kernel applies Adam, one RCCL all-reduce exchanges gradients under data parallel;
  * each update (G step, D step) is captured once into a hipGraph and replayed: ~250 kernel launches
    per update cost one graph launch.  The RNG is counter-based with device-side state, so replays
    draw fresh noise / labels / dequantisation.
"""
import contextlib
import gc
import math

import numpy as np
import torch
import torch.distributed as _dist

from .. import functional as Fn
from .. import kernels as K
from .. import parallel
from ..common import resnet_block as blocks
from ..common.ops import embedding as _embedding
from ..common.ops import linear as _linear
from ..common.ops import conv2d as _conv2d
from ..common.ops import normalization as _normalization
from ..common.ops import sn as _sn
from ..common.ops.sn import NO_OPS
from ..store import ParamStore, get_default_store, set_default_store

BATCH_SIZE = 64  # Critic batch size                                      (:38)
GEN_BS_MULTIPLE = 2  # Generator batch size, as a multiple of BATCH_SIZE  (:39)
ITERS = 100000
DIM_G = 128
DIM_D = 128
NORMALIZATION_G = True
NORMALIZATION_D = False
OUTPUT_DIM = 3072
LR = 0.0002
DECAY = True
N_CRITIC = 5
CONDITIONAL = True
ACGAN = False
VOCAB_SIZE = 10
EMBEDDING_DIM = 300
LOSS_TYPE = 'HINGE'
SOFT_PLUS = False        # (:63) the softplus-wrapped variants of the three losses (:364-386, :483-497)
N_TOWERS = 2  # len(DEVICES) after the single-GPU hack (:73-75)

nonlinearity = blocks.nonlinearity
Normalize = blocks.Normalize
ConvMeanPool = blocks.ConvMeanPool
MeanPoolConv = blocks.MeanPoolConv
UpsampleConv = blocks.UpsampleConv
ResidualBlock = blocks.ResidualBlock
OptimizedResBlockDisc1 = blocks.OptimizedResBlockDisc1


GEN_FEED_ONE_LAUNCH = True      # the label draw, the noise draw and the statistics arena's fill in front of a generator pass as one launch


def generator_arena_floats(groups):
    """floats of the statistics arena of one generator pass (0: the pass takes none)"""
    return 6 * groups * 16 * 2 * DIM_G * 2 + 512 if (NORMALIZATION_G and Fn.CONV_EPILOGUE_STATS) else 0


def Generator(n_samples_, labels, noise=None, reuse=False, groups=1, rng_state=None, arena_buf=None, out=None):
    """(:237-263)  noise [n,128] bf16 (drawn from the device RNG when None) -> [n, 3072] bf16, HWC order,
    tanh range.  `groups` towers of n/groups samples have independent CBN statistics.  arena_buf: the pass's statistics arena,
    cleared by the caller (kernels.generator_feed).  out: a bf16 buffer of n * 3072 elements a no-grad pass writes its result into
    (the fused G.OutputNorm + G.Output launch only; other paths ignore it -- the caller compares data pointers)."""
    store = get_default_store()
    # the statistics sums of the six convs that feed a conditional batch norm: one fill for the whole pass
    arena = K.stats_arena(generator_arena_floats(groups), labels.device, arena_buf) if (generator_arena_floats(groups) and labels.is_cuda) \
        else contextlib.nullcontext()
    with store.variable_scope("Generator", reuse=reuse), arena:
        if noise is None:
            if rng_state is None:
                raise ValueError('Generator needs `noise` or an `rng_state` (kernels.new_rng_state)')
            noise = K.rng_normal((n_samples_, 128), rng_state)                 # tf.random_normal (:240)
        output = _linear.Linear(noise, 128, 4 * 4 * DIM_G * 8, 'G.Input')
        output = output.reshape(-1, 4, 4, DIM_G * 8)
        output = ResidualBlock(output, DIM_G * 8, DIM_G * 2, 3, 'G.Block.1', resample='up', labels=labels, biases=True, groups=groups, out_stats=groups)
        output = Fn.boundary(output, 'G.Block.1')      # gradient-bucket boundaries of the data-parallel backward pass
        output = ResidualBlock(output, DIM_G * 2, DIM_G * 2, 3, 'G.Block.2', resample='up', labels=labels, biases=True, groups=groups, out_stats=groups)
        output = Fn.boundary(output, 'G.Block.2')
        output = ResidualBlock(output, DIM_G * 2, DIM_G * 2, 3, 'G.Block.3', resample='up', labels=labels, biases=True, groups=groups, out_stats=groups)
        output = Fn.boundary(output, 'G.Block.3')
        if (FUSE_OUTPUT_NORM and not torch.is_grad_enabled() and NORMALIZATION_G and blocks.CONDITIONAL and labels is not None
                and output.is_cuda):
            # a pass that keeps nothing for a backward pass (the critic's fakes, sampling): G.OutputNorm + relu ride on
            # G.Output's operand staging -- the normalised 32x32x256 tensor (168 MB at 320 samples) is never written or read
            with store.variable_scope('G.OutputNorm'):
                gamma, beta = _normalization.cond_batchnorm_variables(DIM_G * 2, 10)
            stats = K.cbn_stats(output, groups, getattr(output, '_cbn_stats', None))
            filters, biases = _conv2d.conv2d_variables(DIM_G * 2, 3, 3, 1, 'G.Output', he_init=False)
            wf, _ = Fn._prepared(filters, 3, DIM_G * 2, 3, True, False)
            output = K.cbn_relu_conv3x3_fprop(output, labels, gamma.detach(), beta.detach(), stats, wf, biases.detach(), 3, K.OUT_TANH, out=out)
            return output.reshape(-1, OUTPUT_DIM)
        output = Normalize('G.OutputNorm', output, labels, groups=groups, relu=True)    # + nonlinearity (:257-258)
        output = _conv2d.Conv2D(output, DIM_G * 2, 3, 3, 1, 'G.Output', he_init=False, out_tanh=True)  # + tanh (:260-261)
        return output.reshape(-1, OUTPUT_DIM)


INCEPTION_FREQUENCY = 1000   # how frequently to calculate the Inception score (:49)
FUSE_OUTPUT_NORM = True    # no-grad passes: G.OutputNorm + relu inside G.Output's operand staging
# Data parallel: the generator's gradient buffer (31.5 MB fp32) leaves in these buckets, last layers first, each as soon
# as the backward pass has passed the block boundary below it (parallel.GradBuckets).  Forward / creation order.
G_BUCKETS = (('G.Input/', 'G.Block.1.'), ('G.Block.2.',), ('G.Block.3.',), ('G.OutputNorm/', 'G.Output/'))
BUCKETED_G_ALLREDUCE = True
BATCH_SMALL_WGRADS = True    # same-shape small filter gradients of an update are issued in one launch


def _d_prep_kind(name, W):
    """MFMA operand layout per critic weight (kernels.prep_weights_batched): the two ConvMeanPool 3x3 layers run as
    4x4 stride-2 convs, the two small dense layers read their fp32 weight directly."""
    if W.dim() == 2 and W.numel() <= 65536:
        return None
    if Fn.POOL_CONV4 and name.endswith(('D.Block.1.Conv2/Filters', 'D.Block.2.Conv2/Filters')):
        if Fn.CPOOL_RESIDENT and W.dim() == 4 and W.shape[3] == 128 and W.shape[2] % 128 == 0:
            return 5           # resident ConvMeanPool kernels (geometry of both critic layers: pooled 16x16 and 8x8)
        return 2
    if blocks.FUSE_RES8 and W.dim() == 4 and tuple(W.shape) == (3, 3, 128, 128) and ('D.Block.3.' in name or 'D.Block.4.' in name):
        return 4                                                          # fused 8x8 residual blocks: "rfrag" operands
    if Fn.IMG16_CONV and name.endswith('D.Block.2.Conv1/Filters') and W.dim() == 4 and W.shape[0] == 3 and W.shape[2] % 64 == 0 and W.shape[3] % 128 == 0 \
            and W.shape[2] % 128 == 0:
        if Fn.FACTOR_LABEL_CONV and LABEL_TABLE and CONDITIONAL and W.shape[2] == 2 * DIM_D:
            return 6                                                      # ... on the feature half; the tiled label half is a bias table
        return 4                                                          # 16x16 image-resident conv (forward and input gradient)
    return 0


def _g_prep_kind(name, W):
    """the UpsampleConv 3x3 layers run phase-decomposed; the first block's two convs (4x4 -> 8x8 and 8x8) take the
    LDS-resident kernel and its fragment-major operands"""
    if Fn.RES8_CONV and name.endswith(('G.Block.1.Conv1/Filters', 'G.Block.1.Conv2/Filters')) and W.dim() == 4 and W.shape[0] == 3 \
            and W.shape[2] in (128, 256) and W.shape[3] % 128 == 0:
        return 4
    if Fn.PHASE_UPCONV and name.endswith('.Conv1/Filters') and W.dim() == 4 and W.shape[0] == 3:
        return 1
    if Fn.IMG16_CONV and name.endswith('G.Block.2.Conv2/Filters') and W.dim() == 4 and W.shape[0] == 3 and W.shape[2] % 128 == 0 and W.shape[3] % 128 == 0:
        return 4                                                          # 16x16 image-resident conv (forward with statistics, input gradient)
    return 0


FUSE_SN_TAIL = True    # critic update: spectral-norm backward apply + TF-Adam + the NEXT pass's power iteration as ONE launch (gank_sn_adam_fwd_a)
FUSE_FEED = True        # the critic's input feed rides on the second spectral-norm launch of its forward pass (gank_sn_power_iter_fwd_b_prep_feed)
LABEL_TABLE = True    # the critic's label branch through a per-label table (0: per-sample embedding + dense layer + tile)
FUSED_HEAD = True      # D.Output + hinge loss (+ the layer's three gradients) as one launch where the train step asks for it
HEAD_IN_CHAIN = True   # ... and that launch folded into the fused 8x8 chain's forward / backward launches (functional.HingeHeadSpec)


def Discriminator(inputs, labels, update_collection=None, reuse=False, loss_head=None):
    """(:266-313, ACGAN=False)  inputs [n,3072] bf16 (HWC order) -> (logits [n], None).
    loss_head: a callable (features [n, DIM_D], W_bar [DIM_D, 1], b [1]) -> loss that replaces the last dense layer AND the
    loss on its logits (functional.hinge_d_head / hinge_g_head): the call then returns (loss, None)."""
    store = get_default_store()
    with store.variable_scope("Discriminator", reuse=reuse):
        prefix = store.full_name('')[:-1]
        # the label branch as a per-label table out of the spectral norm's second launch (functional.concat_label)
        label_dense = None
        if LABEL_TABLE:
            names = [prefix + '/Embedding.Label/embedding_map', prefix + '/D.Embedding_y/W', prefix + '/D.Embedding_y/b']
            if all(nm in store.vars for nm in names):
                label_dense = (store.vars[names[0]], names[1], store.vars[names[2]])
        with _sn.precomputed(store, prefix, update_collection, prep_kind=_d_prep_kind, label_dense=label_dense):   # one batched SN for all 12 weights
            output = inputs.reshape(-1, 32, 32, 3)
            prefork = None
            if LABEL_TABLE:
                # embed_y -> Linear -> expand_dims x2 -> tile -> concat (:276-284) depends on a sample only through its label:
                # the dense layer runs on the 10 table rows and the concat gathers (same variables, same creation order)
                emb_table = _embedding.embedding_variable(VOCAB_SIZE, EMBEDDING_DIM)
                w_emb, b_emb = _linear.linear_variables(EMBEDDING_DIM, DIM_D, 'D.Embedding_y', spectral_normed=True,
                                                        update_collection=update_collection, biases=True)
                output = OptimizedResBlockDisc1(output, spectral_normed=True, update_collection=update_collection, biases=True)
                if (blocks.FUSE_LABEL_FORK and blocks.COMMUTE_1X1 and blocks.FUSE_FORK_POOL and output.requires_grad
                        and output.shape[1] % 2 == 0 and output.shape[2] % 2 == 0 and 1024 % ((output.shape[3] + DIM_D) // 8) == 0):
                    w1 = None
                    if Fn.FACTOR_LABEL_CONV:
                        # D.Block.2's variables in ResidualBlock's order (shortcut, conv_1; conv_2 below), then the block itself with conv_1
                        # computed on the feature half alone (the tiled half of its input is one vector per sample)
                        ws, bs = _conv2d.conv2d_variables(DIM_D * 2, DIM_D, 1, 1, 'D.Block.2.Shortcut', spectral_normed=True,
                                                          update_collection=update_collection, he_init=False, biases=True)
                        w1, b1 = _conv2d.conv2d_variables(DIM_D * 2, DIM_D * 2, 3, 1, 'D.Block.2.Conv1', spectral_normed=True,
                                                          update_collection=update_collection, he_init=True, biases=True)
                    if w1 is not None and Fn.concat_label_conv1_ok(output, w1):
                        h1, pooled = Fn.concat_label_conv1(output, labels, emb_table, w_emb, b_emb, w1, b1, ws, bs)
                        shortcut = _conv2d.Conv2D(pooled, DIM_D * 2, DIM_D, 1, 1, 'D.Block.2.Shortcut', spectral_normed=True,
                                                  update_collection=update_collection, he_init=False, biases=True)
                        output = blocks.ConvMeanPool(h1, output_dim=DIM_D, filter_size=3, name='D.Block.2.Conv2', spectral_normed=True,
                                                     update_collection=update_collection, he_init=True, biases=True, in_relu=True,
                                                     residual=shortcut)
                        prefork = 'done'
                    else:
                        prefork = Fn.concat_label_fork_pool(output, labels, emb_table, w_emb, b_emb)     # concat + D.Block.2's fan-out
                        output = None
                else:
                    output = Fn.concat_label(output, labels, emb_table, w_emb, b_emb)
            else:
                embedding_y = _embedding.embed_y(labels, VOCAB_SIZE, EMBEDDING_DIM)       # label embedding -> dense layer (:279-281)
                embedding_y = _linear.Linear(embedding_y, EMBEDDING_DIM, DIM_D, 'D.Embedding_y', spectral_normed=True,
                                             update_collection=update_collection, biases=True)
                output = OptimizedResBlockDisc1(output, spectral_normed=True, update_collection=update_collection, biases=True)
                output = Fn.concat_tile(output, embedding_y)               # expand_dims x2 + tile + concat (:282-284)
            if prefork != 'done':
                output = ResidualBlock(output, DIM_D * 2, DIM_D, 3, 'D.Block.2', spectral_normed=True,
                                       update_collection=update_collection, resample='down', labels=labels, biases=True, prefork=prefork)
            if blocks.res_chain8_eligible(output, DIM_D, ['D.Block.3', 'D.Block.4'], labels):
                # D.Block.3, D.Block.4 and nonlinearity + reduce_mean (:291-301) as ONE launch: an 8x8x128 sample stays in LDS
                if HEAD_IN_CHAIN and isinstance(loss_head, Fn.HingeHeadSpec) and not (CONDITIONAL and ACGAN):
                    # ... and D.Output + the hinge loss (:303-304, :379-381 / :492) inside the same two launches
                    return blocks.ResidualBlockChain8(
                        output, DIM_D, ['D.Block.3', 'D.Block.4'], spectral_normed=True, update_collection=update_collection, biases=True, pool=True,
                        head=(loss_head, lambda: _linear.linear_variables(DIM_D, 1, 'D.Output', spectral_normed=True,
                                                                          update_collection=update_collection))), None
                output = blocks.ResidualBlockChain8(output, DIM_D, ['D.Block.3', 'D.Block.4'], spectral_normed=True,
                                                    update_collection=update_collection, biases=True, pool=True)
            else:
                output = ResidualBlock(output, DIM_D, DIM_D, 3, 'D.Block.3', spectral_normed=True,
                                       update_collection=update_collection, resample=None, labels=labels, biases=True)
                output = ResidualBlock(output, DIM_D, DIM_D, 3, 'D.Block.4', spectral_normed=True,
                                       update_collection=update_collection, resample=None, labels=labels, biases=True)
                output = Fn.relu_meanpool_hw(output)                       # nonlinearity + reduce_mean (:299-301)
            if CONDITIONAL and ACGAN:            # two heads read the pooled features (:302-311)
                out_a, out_b = Fn.fork(output)
                output_wgan = _linear.Linear(out_a, DIM_D, 1, 'D.Output', spectral_normed=True,
                                             update_collection=update_collection)
                output_acgan = _linear.Linear(out_b, DIM_D, 10, 'D.ACGANOutput', spectral_normed=True,
                                              update_collection=update_collection, biases=True)
                return output_wgan.reshape(-1), output_acgan
            if loss_head is not None:
                w_out, b_out = _linear.linear_variables(DIM_D, 1, 'D.Output', spectral_normed=True, update_collection=update_collection)
                return loss_head(output, w_out, b_out), None
            output_wgan = _linear.Linear(output, DIM_D, 1, 'D.Output', spectral_normed=True,
                                         update_collection=update_collection)
            return output_wgan.reshape(-1), None


def lr_decay(iteration):
    """(:454-459)"""
    if not DECAY:
        return 1.
    return max(0., 1. - iteration / 100000.) if iteration < 50000 else 0.5


class AdamTF:
    """tf.train.AdamOptimizer(beta1=0., beta2=0.9) over one flat buffer (:521-526)."""

    def __init__(self, flat, iteration, lr=LR, beta1=0., beta2=0.9, eps=1e-8, grad_scale=1.0, decay=DECAY, health=False):
        """All step state is device resident (hyper-parameters, step count t, the shared `iteration`
        counter that drives the LR decay), so a captured update replays with no host traffic."""
        self.flat, self.iteration = flat, iteration
        dev = flat["params"].device
        self.hp = torch.tensor([lr, beta1, beta2, eps, grad_scale, 1.0 if decay else 0.0, 0.0, 0.0],
                               dtype=torch.float32, device=dev)
        self.t = torch.zeros(1, dtype=torch.int64, device=dev)
        # {non-finite gradients, zero gradients} seen so far (loss-scaled runs: overflow / underflow watch); None = not counted
        self.health = torch.zeros(2, dtype=torch.int64, device=dev) if health else None
        self.pending_sn = None       # a batched spectral norm whose backward apply this optimiser's next launch performs (functional.defer_sn_apply)
        self.sn_state = None         # the network's persistent spectral-norm state, if any: a plain step invalidates it
        self.bump = None             # (counter int64[1], condition int32[1]): the fused launch advances the counter when the condition word is 0
        self.bumped = False          # ... and says so here (train_iteration then skips its own counter launch)

    def apply(self):
        """One launch: the update, the step count, and the gradient buffer (with its scratch half) cleared for the next
        backward pass -- `flat['clean']` tells the trainer that no fill is needed.  With a deferred spectral-norm backward
        pending, the launch is gank_sn_adam_fwd_a: that backward's apply step, this update, and the next forward pass's power
        iteration on the updated weights."""
        f = self.flat
        batch, self.pending_sn = self.pending_sn, None
        if batch is not None:
            # (dw_zero: a spectrally normalised weight receives gradient through its normalised copy only, and this launch's
            #  predecessor cleared the buffer: the weights' own gradient views are zero and are not touched)
            bump = self.bump if self.bump is not None else (None, None)
            batch.adam_fwd_a(f["params"], f["grads"], f["m"], f["v"], self.hp, self.t, self.iteration, health=self.health, dw_zero=True,
                             bump=bump[0], bump_when_zero=bump[1])
            self.bumped = self.bump is not None
        else:
            self.bumped = False
            if self.sn_state is not None:
                self.sn_state.valid = False
            K.adam_tf(f["params"], f["grads_all"], f["m"], f["v"], self.hp, self.t, self.iteration, zero_grads=True, health=self.health)
        f["clean"] = True


@contextlib.contextmanager
def _capture(graph):
    """hipGraph capture with the Python garbage collector paused: a cyclic-GC pass in the middle of a capture can
    destroy an older trainer's CUDAGraph / return its pool memory while the stream is capturing, which aborts the
    process (torch only collects once, on entry)."""
    was = gc.isenabled()
    parallel.drain_collective_watchdog()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        gc.disable()
        try:
            yield
        finally:
            if was:
                gc.enable()


class SNGANTrainer:
    """One process = one GPU.  `world_size` > 1: each rank holds a full replica and its own
    BATCH_SIZE-sample shard of the real data (weak scaling) or BATCH_SIZE/world_size (strong);
    G and D gradients are summed with one RCCL all-reduce per update over the flat gradient buffer
    and averaged inside the Adam kernel (grad_scale = 1/world_size; the reference averages its tower
    losses, :436,:498)."""

    def __init__(self, batch_size=BATCH_SIZE, device="cuda", seed=0, use_graphs=True, process_group=None, state=None,
                 allow_eager_fallback=False, capture_collectives=None, grad_wire_dtype=None, loss_scale=None, loss_type=None, soft_plus=None):
        """allow_eager_fallback: a failed hipGraph capture degrades to eager execution (with a message on stderr) instead of
        raising -- a run that asked for graphs never silently becomes a 10x slower eager run otherwise.
        capture_collectives: under data parallel the RCCL all-reduces are captured INSIDE the update graphs (one graph per
        critic update, one for the whole bucketed generator update) instead of being issued eagerly between graph replays;
        a capture that fails falls back to the split form (graph / eager collective / graph) with a message.  Default (None):
        on for a world-size-1 group (rehearsals: the replayed graph then holds NO collective node, RCCL's single-rank
        all-reduce is a no-op), OFF for world > 1 -- a replayed RCCL collective has never run on more than one GPU in this
        build's history (no multi-GPU node was available), so the multi-rank default is the well-trodden split form until a
        run has verified replay against the eager exchange; pass True to opt in.  The ranks agree on the setting (MIN).
        grad_wire_dtype: 'bf16' sends the gradient buckets over xGMI in the 16-bit activation dtype (half the bytes).
        loss_scale: static loss scale, a power of two (default: 1 for bfloat16 buffers, 1024 for the fp16 build, whose activation
        gradients would otherwise underflow: hinge d loss / d logit is +-1/n and shrinks from there).  The loss nodes multiply
        d loss / d logits by it, the optimisers divide it out (grad_scale) and count non-finite / zero gradients (`health()`)."""
        self.device = torch.device(device)
        self.allow_eager_fallback = allow_eager_fallback
        # LOSS_TYPE of the script (:62, SOFT_PLUS = False :63): 'HINGE' (:371-381, :487-492), 'Goodfellow' (-mean(log sigmoid(real)) -
        # mean(log(1 - sigmoid(fake))); generator -mean(log sigmoid(fake)): :363-369, :483-486), 'WGAN' (mean(fake) - mean(real);
        # generator -mean(fake): :382-387, :493-497)
        self.loss_type = LOSS_TYPE if loss_type is None else loss_type
        # SOFT_PLUS (:63): critic -softplus(log sigmoid(real)) - softplus(log(1 - sigmoid(fake))) | softplus(-min(0, -1 + real)) +
        # softplus(-min(0, -1 - fake)) | softplus(fake) + softplus(-real); generator softplus(-log sigmoid(fake)) | softplus(-fake) (both)
        self.soft_plus = SOFT_PLUS if soft_plus is None else bool(soft_plus)
        if self.loss_type not in ('HINGE', 'Goodfellow', 'WGAN'):
            raise NotImplementedError("LOSS_TYPE %r (gan_cifar_resnet.py:62 knows 'Goodfellow', 'HINGE', 'WGAN'; 'WGAN-GP' has no branch in the script)" % (self.loss_type,))
        self.batch = batch_size
        self.store = set_default_store(ParamStore(self.device, seed=seed))
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        self.use_graphs = use_graphs
        if loss_scale is None:
            loss_scale = 1024.0 if K.BF16 is torch.float16 else 1.0
        self.loss_scale = float(loss_scale)
        assert self.loss_scale > 0 and math.log2(self.loss_scale) == int(math.log2(self.loss_scale)), "loss_scale must be a power of two"
        self.dp = process_group is not None          # the data-parallel path (also for a world-size-1 group: rehearsal / --force-dp)
        # only RCCL collectives are stream-ordered device work that a hipGraph can hold; gloo (the CPU rehearsal backend)
        # synchronises the stream from the host, which a capture must never see
        if capture_collectives is None:
            capture_collectives = self.world == 1
        self.capture_collectives = bool(capture_collectives and process_group is not None
                                        and _dist.get_backend(process_group) == "nccl")
        self.grad_wire_dtype = K.BF16 if grad_wire_dtype in ('bf16', 'fp16', '16') else None
        self._g_buckets = None
        # world > 1: the generator's gradients leave in buckets beside the backward pass; `bucketed` forces that path for a
        # single rank too (tests: same arithmetic as the one-piece update)
        self.bucketed = BUCKETED_G_ALLREDUCE and process_group is not None
        # data-side RNG differs per rank; parameter init (store seed) is identical on all ranks
        self.rng_state = K.new_rng_state(parallel.data_seed(seed, self.rank), self.device)
        self.iteration = 0
        self._build(state)
        self._graphs = {}
        self._g_applier = type("_Apply", (), {"apply": staticmethod(self._g_apply)})()
        # tests: callable(stage, k) invoked by train_iteration between its pieces ('g', 'gen5', 'd_pre' before / after each),
        # so the captured path can be inspected where bench.py times it; None in production (no host work is added)
        self.observer = None
        self.last_logits = None      # the critic logits of the latest prefetched update (a graph-pool tensor under replay)

    # ---- graph construction (variables are created by name on the first call, :238,:267) ----------
    def _build(self, state):
        b = self.batch
        with torch.no_grad():
            labels = torch.zeros(b, dtype=torch.int32, device=self.device)
            z = torch.zeros((b, 128), dtype=K.BF16, device=self.device)
            fake = Generator(b, labels, noise=z, groups=N_TOWERS)
            Discriminator(fake, labels, update_collection=NO_OPS)
        if state is not None:
            self.store.load_state_dict(state)
        self.g_flat = self.store.flatten('Generator')
        self.d_flat = self.store.flatten('Discriminator', scratch_tail=True)
        self.store.flatten_state('Discriminator')      # the 12 SN u vectors: one buffer
        self.sn_state = None
        if FUSE_SN_TAIL and self.device.type == 'cuda':
            # persistent spectral-norm workspaces: the critic's optimiser launch runs the NEXT forward pass's power iteration
            pairs = _sn.sn_pairs(self.store, 'Discriminator')
            u_flat = _sn._flat_base(self.store, 'Discriminator', [u for _, u in pairs])
            if pairs and len(pairs) <= 16 and u_flat is not None and all(w.shape[-1] <= 256 for w, _ in pairs):
                self.sn_state = self.store.sn_state['Discriminator'] = K.SnState([w for w, _ in pairs], [u for _, u in pairs], u_flat)
        # bf16 MFMA operand copies of the generator weights: rebuilt by ONE launch after each G update
        # (the generator runs 6 forwards per iteration on unchanged weights)
        self._g_convs = [(k, v) for k, v in self.store.vars.items()
                         if k.startswith('Generator/') and (k.endswith('/Filters') or k.endswith('/W'))]
        self._refresh_g_prep()
        self.iteration_dev = torch.zeros(1, dtype=torch.int64, device=self.device)   # `_iteration` feed (:320)
        scaled = self.loss_scale != 1.0
        self.g_opt = AdamTF(self.g_flat, self.iteration_dev, grad_scale=1.0 / (self.world * self.loss_scale), health=scaled)
        self.d_opt = AdamTF(self.d_flat, self.iteration_dev, grad_scale=1.0 / (self.world * self.loss_scale), health=scaled)
        self.d_opt.sn_state = self.sn_state
        # static input buffers (graph replays read these addresses)
        self.real_u8 = torch.zeros((b, OUTPUT_DIM), dtype=torch.uint8, device=self.device)
        self.real_labels = torch.zeros(b, dtype=torch.int32, device=self.device)
        # one iteration's worth of critic feeds and generator outputs (train_iteration)
        self.real_all = torch.zeros((N_CRITIC, b, OUTPUT_DIM), dtype=torch.uint8, device=self.device)
        self.labels_all = torch.zeros((N_CRITIC, b), dtype=torch.int32, device=self.device)
        self.fake_all = torch.zeros((N_CRITIC, b, OUTPUT_DIM), dtype=K.BF16, device=self.device)
        # the critic's input of one update, laid out by ONE launch from slot `feed_slot` of the ring above
        self.both = torch.zeros((2 * b, OUTPUT_DIM), dtype=K.BF16, device=self.device)
        self.both_labels = torch.zeros(2 * b, dtype=torch.int32, device=self.device)
        self.feed_slot = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.feed_done = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._seed = {}
        self.d_loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.g_loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _refresh_g_prep(self):
        K.prep_weights_batched([v for _, v in self._g_convs], want_d=True, kinds=[_g_prep_kind(k, v) for k, v in self._g_convs])

    # ---- checkpoint (tf.train.Saver of the reference, :585-590: variables, Adam slots, beta powers) ------------
    def state_dict(self):
        """Everything a resumed run needs to continue the SAME trajectory: the named variables (TF names), the Adam
        slots as `<var>/Adam` (m) and `<var>/Adam_1` (v) -- the names tf.train.AdamOptimizer gives them --, the two
        optimisers' step counts (TF stores them as beta1_power / beta2_power; `<net>/adam_t` here, from which the
        powers follow), the iteration counter that drives the LR decay (:454-459), and the device RNG state."""
        sd = self.store.state_dict()
        for net, flat, opt in (('Generator', self.g_flat, self.g_opt), ('Discriminator', self.d_flat, self.d_opt)):
            m, v = flat['m'].detach().cpu().numpy(), flat['v'].detach().cpu().numpy()
            for k in flat['names']:
                o, n = flat['offsets'][k], self.store.vars[k].numel()
                shape = tuple(self.store.vars[k].shape)
                sd[k + '/Adam'] = m[o:o + n].reshape(shape).copy()
                sd[k + '/Adam_1'] = v[o:o + n].reshape(shape).copy()
            sd[net + '/adam_t'] = np.asarray(int(opt.t.item()), dtype=np.int64)
        sd['_iteration'] = np.asarray(int(self.iteration), dtype=np.int64)
        sd['_rng_state'] = self.rng_state.detach().cpu().numpy().copy()
        return sd

    def load_state_dict(self, state, strict=True):
        """Restore variables by name; optimiser slots / step counts / iteration / RNG state are restored when the
        checkpoint carries them and RESET otherwise (a weights-only checkpoint restarts Adam's bias correction, as a
        fresh tf.train.AdamOptimizer would).  Cached operand copies and captured graphs are rebuilt."""
        state = dict(state)
        extra = {k: state.pop(k) for k in list(state) if k.endswith(('/Adam', '/Adam_1', '/adam_t')) or k in ('_iteration', '_rng_state', '_rng_state_gen')}
        self.store.load_state_dict(state, strict)
        with torch.no_grad():
            for net, flat, opt in (('Generator', self.g_flat, self.g_opt), ('Discriminator', self.d_flat, self.d_opt)):
                flat['m'].zero_()
                flat['v'].zero_()
                for k in flat['names']:
                    o, n = flat['offsets'][k], self.store.vars[k].numel()
                    for suffix, buf in (('/Adam', flat['m']), ('/Adam_1', flat['v'])):
                        val = extra.get(k + suffix)
                        if val is not None:
                            buf[o:o + n].copy_(torch.from_numpy(np.asarray(val, np.float32).reshape(-1)).to(self.device))
                opt.t.fill_(int(extra.get(net + '/adam_t', 0)))
            self.iteration = int(extra.get('_iteration', 0))
            self.iteration_dev.fill_(self.iteration)
            if '_rng_state' in extra:
                self.rng_state.copy_(torch.from_numpy(np.asarray(extra['_rng_state'], np.int64)).to(self.device))
            self.feed_slot.zero_()
        self._refresh_g_prep()
        self._graphs.clear()

    def sn_state_changed(self):
        """Call after writing critic weights or u vectors behind the trainer's back (tests, tools): the power iteration the last
        critic update ran ahead of time (kernels.SnState) no longer belongs to them."""
        if self.sn_state is not None:
            self.sn_state.valid = False

    def _ensure_sn_state(self):
        """Before replaying a captured update that contains a critic pass: a graph captured while the power iteration was already
        done holds only the second spectral-norm launch, so the persistent state must be current -- after an eager pass that
        assigned u without an update behind it (the dev-loss evaluation), or after sn_state_changed(), it is recomputed here."""
        if self.sn_state is not None and not self.sn_state.valid:
            self.sn_state.refresh()

    def health(self):
        """{'G': (non-finite, zero), 'D': (...)} gradient counts since the trainer was built (loss-scaled runs only; else None)"""
        if self.g_opt.health is None:
            return None
        return {'G': tuple(int(v) for v in self.g_opt.health.tolist()), 'D': tuple(int(v) for v in self.d_opt.health.tolist())}

    def _g_apply(self):
        self.g_opt.apply()
        self._refresh_g_prep()

    def _begin_grads(self, flat):
        """Gradients accumulate into the flat buffer, which must be zero when a backward pass starts.  The optimiser launch
        leaves it zero (AdamTF.apply); only a pass that follows another pass without an update in between -- tests and
        tools calling the *_forward_backward pieces directly -- needs the fill."""
        if not flat.get("clean", False):
            flat["grads_all"].zero_()
        flat["clean"] = False

    def _ensure_clean(self, flat):
        """before replaying a captured update (its graph contains no fill)"""
        if not flat.get("clean", False):
            flat["grads_all"].zero_()
            flat["clean"] = True

    # ---- the losses of the script's LOSS_TYPE switch ----------------------------------------------------------
    def _critic_loss(self, both, both_labels, n_real):
        """-> (loss, logits): disc_cost on concat(real, fake) with update_collection=None"""
        if self.loss_type == 'HINGE' and FUSED_HEAD and not self.soft_plus:
            loss, _ = Discriminator(both, both_labels, update_collection=None,
                                    loss_head=Fn.HingeHeadSpec(0, n_real, out=self.d_loss, loss_scale=self.loss_scale))
            return loss, loss.logits
        logits, _ = Discriminator(both, both_labels, update_collection=None)
        if self.soft_plus:               # (:366-367, :372-373, :383-385; softplus(fake) + softplus(-real) IS the sigmoid cross-entropy)
            kind = {'Goodfellow': 5, 'HINGE': 7, 'WGAN': 2}[self.loss_type]
            return Fn.gan_pointwise_loss(logits, n_real, kind, out=self.d_loss), logits
        if self.loss_type == 'HINGE':
            return Fn.hinge_d_loss(logits, n_real, out=self.d_loss), logits
        if self.loss_type == 'WGAN':
            return Fn.wgan_d_loss(logits, n_real, out=self.d_loss), logits
        return Fn.gan_pointwise_loss(logits, n_real, 2, out=self.d_loss), logits          # Goodfellow: sigmoid cross-entropy against ones / zeros

    def _generator_loss(self, fake, fake_labels):
        """-> (loss, logits): gen_cost, critic with update_collection=NO_OPS"""
        if self.loss_type == 'HINGE' and FUSED_HEAD and not self.soft_plus:
            loss, _ = Discriminator(fake, fake_labels, update_collection=NO_OPS,
                                    loss_head=Fn.HingeHeadSpec(1, 0, out=self.g_loss, loss_scale=self.loss_scale))
            return loss, loss.logits
        logits, _ = Discriminator(fake, fake_labels, update_collection=NO_OPS)
        if self.soft_plus:               # (:484, :489-490, :494-495)
            return Fn.gan_pointwise_loss(logits, 0, 6 if self.loss_type == 'Goodfellow' else 3, out=self.g_loss), logits
        if self.loss_type in ('HINGE', 'WGAN'):
            return Fn.hinge_g_loss(logits, out=self.g_loss), logits                         # -mean(disc_fake) in both branches
        return Fn.gan_pointwise_loss(logits, 0, 3, out=self.g_loss), logits                 # Goodfellow: -mean(log sigmoid(disc_fake))

    # ---- the two updates, as plain eager code (captured into graphs by _run) -----------------------
    def _d_forward_backward(self, real_pre=None, z=None, fake=None):
        """disc_cost and its gradients (:326-381): fakes from N_TOWERS generator towers conditioned on
        the REAL labels, critic on concat(real, fake) with update_collection=None.  `fake`: precomputed
        generator output for this update (see _generate_for_critic)."""
        set_default_store(self.store)
        b = self.batch
        self._begin_grads(self.d_flat)
        with torch.no_grad():   # generator is not trained by disc_cost: no autograd graph through it
            if fake is None:
                fake = Generator(b, self.real_labels, noise=z, groups=N_TOWERS, rng_state=self.rng_state)
            real = K.preprocess_real(self.real_u8, self.rng_state).reshape(b, OUTPUT_DIM) if real_pre is None else real_pre
            both = torch.cat([real, fake], 0)                          # plumbing: device memcpy
            both_labels = torch.cat([self.real_labels, self.real_labels], 0)
        with _sn.grad_scratch(self.d_flat["scratch"]):        # zeroed by zero_grads above
            loss, logits = self._critic_loss(both, both_labels, b)
        self._backward(loss)
        return logits

    def _d_forward_backward_prefetched(self):
        """Critic update number `feed_slot` of the iteration: inputs come from the feed ring by one launch."""
        set_default_store(self.store)
        self._begin_grads(self.d_flat)
        feed = (self.real_all, self.labels_all, self.fake_all, self.both, self.both_labels, self.feed_slot, self.rng_state, self.feed_done)
        if FUSE_FEED:
            K.defer_critic_feed(*feed)        # launched by the spectral-norm forward pass in front of the first layer
        else:
            K.critic_feed(*feed)
        with _sn.grad_scratch(self.d_flat["scratch"]):        # zeroed by zero_grads above
            loss, logits = self._critic_loss(self.both, self.both_labels, self.batch)
        if K.deferred_critic_feed_pending():
            raise RuntimeError("the critic's forward pass launched no spectral-norm batch: its deferred feed never ran")
        self._backward(loss)
        self.last_logits = logits
        return logits

    @torch.no_grad()
    def _generate_for_critic(self):
        """The generator is frozen during the N_CRITIC critic updates of an iteration (:599-620), so their
        fakes are ONE generator pass over N_CRITIC*B samples: each update's N_TOWERS towers keep their own
        conditional-batch-norm statistics (`groups`), each sample its own noise -- the same arithmetic as
        N_CRITIC separate passes, in 5x fewer, 5x larger launches."""
        set_default_store(self.store)
        n = N_CRITIC * self.batch
        z = abuf = None
        if GEN_FEED_ONE_LAUNCH and self.labels_all.is_cuda:
            _, z, abuf = K.generator_feed(self.rng_state, (n, 128), generator_arena_floats(N_CRITIC * N_TOWERS))
        fake = Generator(n, self.labels_all.reshape(-1), noise=z, groups=N_CRITIC * N_TOWERS, rng_state=self.rng_state, arena_buf=abuf,
                         out=self.fake_all if (self.fake_all.is_contiguous() and self.fake_all.numel() == n * OUTPUT_DIM) else None)
        if fake.data_ptr() != self.fake_all.data_ptr():      # (the fused output launch wrote the ring itself: no copy)
            K.copy_(self.fake_all, fake)

    def _g_forward_backward(self, z=None, fake_labels=None):
        """gen_cost and its gradients (:464-498): N_TOWERS towers of GEN_BS_MULTIPLE*B/N_TOWERS samples,
        critic with update_collection=NO_OPS (u read, never written)."""
        set_default_store(self.store)   # the store is the "default graph": several trainers may coexist
        n = GEN_BS_MULTIPLE * self.batch
        self._begin_grads(self.g_flat)
        abuf = None
        if fake_labels is None and z is None and GEN_FEED_ONE_LAUNCH and self.rng_state.is_cuda:
            fake_labels, z, abuf = K.generator_feed(self.rng_state, (n, 128), generator_arena_floats(N_TOWERS), n, 10)       # :467, :240
        if fake_labels is None:
            fake_labels = K.rng_labels(n, 10, self.rng_state)           # :467
        fake = Generator(n, fake_labels, noise=z, groups=N_TOWERS, rng_state=self.rng_state, arena_buf=abuf)
        d_params = self.store.params_with_name('Discriminator')
        for p in d_params:      # gen_cost is differentiated w.r.t. gen_params only (:523)
            p.requires_grad_(False)
        try:
            loss, logits = self._generator_loss(fake, fake_labels)
            self._backward(loss)
        finally:
            for p in d_params:
                p.requires_grad_(True)
        return logits

    def _backward(self, loss):
        """loss.backward() with the small filter gradients deferred and issued in batches, before anything reads them."""
        Fn.reset_deferred()
        Fn.BATCH_SMALL_WGRADS = BATCH_SMALL_WGRADS
        try:
            # the gradient seed is a persistent tensor: loss.backward() alone launches a ones_like fill per update
            loss.backward(gradient=Fn.grad_seed(loss, self.loss_scale))
            Fn.join_wgrad()
        finally:
            Fn.reset_deferred()      # empty after a clean join; after an exception: nothing stale survives
            Fn.BATCH_SMALL_WGRADS = False

    # ---- data parallel generator update: segmented backward, bucketed all-reduce beside it ---------------------------
    def _g_phases(self):
        """The generator update as a list of phases [forward, backward segment 0 .. 3, optimiser] and, per phase, the
        bucket whose all-reduce may start when that phase has been enqueued (None: nothing).  Segment k runs the backward
        pass from the boundary above bucket k down to the boundary below it."""
        st = {}
        nb = len(G_BUCKETS)

        def forward():
            set_default_store(self.store)
            n = GEN_BS_MULTIPLE * self.batch
            self._begin_grads(self.g_flat)
            fake_labels = K.rng_labels(n, 10, self.rng_state)
            with Fn.record_boundaries() as marks:
                fake = Generator(n, fake_labels, groups=N_TOWERS, rng_state=self.rng_state)
            d_params = self.store.params_with_name('Discriminator')
            for p in d_params:
                p.requires_grad_(False)
            try:
                loss, _ = self._generator_loss(fake, fake_labels)
            finally:
                for p in d_params:
                    p.requires_grad_(True)
            cuts = [t for tag, t in marks if tag in ('G.Block.1', 'G.Block.2', 'G.Block.3')]
            assert len(cuts) == nb - 1, [tag for tag, _ in marks]
            st['top'], st['gtop'], st['cuts'] = loss, Fn.grad_seed(loss, self.loss_scale), cuts

        def segment(k):          # k = nb-1 (next to the loss) .. 0 (the network input side)
            def run():
                names = [nm for nm in self.g_flat['names'] if any(sub in nm for sub in G_BUCKETS[k])]
                params = [self.store.vars[nm] for nm in names]
                below = st['cuts'][k - 1] if k > 0 else None
                inputs = ([below] if below is not None else []) + params
                Fn.reset_deferred()
                Fn.BATCH_SMALL_WGRADS = BATCH_SMALL_WGRADS
                try:
                    grads = torch.autograd.grad([st['top']], inputs, [st['gtop']], allow_unused=True)
                    Fn.join_wgrad()
                finally:
                    Fn.reset_deferred()
                    Fn.BATCH_SMALL_WGRADS = False
                if below is not None:
                    st['top'], st['gtop'] = below, grads[0]
                else:
                    st.clear()
            return run

        phases = [forward] + [segment(k) for k in reversed(range(nb))] + [self._g_apply]
        after = [None] + list(reversed(range(nb))) + [None]
        return phases, after

    def _g_step_bucketed(self):
        if self._g_buckets is None:
            self._g_buckets = parallel.GradBuckets(self.g_flat['grads'], parallel.bucket_ranges(self.g_flat, G_BUCKETS), self.pg, self.grad_wire_dtype)
        gb = self._g_buckets
        phases, after = self._g_phases()
        last = len(phases) - 1

        def between(i):
            if after[i] is not None:
                gb.launch(after[i])
            if i == last - 1:
                gb.join()              # the optimiser phase reads every bucket

        if not self.use_graphs:
            for i, ph in enumerate(phases):
                ph()
                between(i)
            return
        if 'g_seg' not in self._graphs:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):          # eager first execution (this IS the update), exactly as _run does
                for i, ph in enumerate(phases):
                    ph()
                    between(i)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            if self.capture_collectives:
                try:
                    g = torch.cuda.CUDAGraph()
                    with _capture(g):           # every phase AND the bucket all-reduces (forked onto the communication stream): ONE graph
                        for i, ph in enumerate(phases):
                            ph()
                            between(i)
                    self._graphs['g_seg'] = g
                except Exception as e:  # noqa: BLE001
                    import sys
                    torch.cuda.synchronize()
                    print(f"[gank] capturing the bucketed generator update with its collectives failed ({e}); one graph per phase, "
                          f"collectives between them", file=sys.stderr)
                    self.capture_collectives = False
                self._agree_on_capture()        # (entered with the same setting on every rank, so every rank calls this)
                if isinstance(self._graphs.get('g_seg'), torch.cuda.CUDAGraph):
                    if self.capture_collectives:
                        return
                    del self._graphs['g_seg']   # another rank could not capture its collectives: every rank takes the split form
            try:
                pool = torch.cuda.graph_pool_handle()
                graphs = []
                for ph in phases:               # one graph per phase, one memory pool: later phases read what earlier ones made
                    g = torch.cuda.CUDAGraph()
                    was = gc.isenabled()
                    parallel.drain_collective_watchdog()
                    with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                        gc.disable()
                        try:
                            ph()
                        finally:
                            if was:
                                gc.enable()
                    graphs.append(g)
                self._graphs['g_seg'] = graphs
            except Exception as e:  # noqa: BLE001
                self._capture_failed('bucketed generator update', e)
            return
        self._ensure_clean(self.g_flat)
        self._ensure_sn_state()
        if isinstance(self._graphs['g_seg'], torch.cuda.CUDAGraph):
            self._graphs['g_seg'].replay()
        else:
            for i, g in enumerate(self._graphs['g_seg']):
                g.replay()
                between(i)
        self.g_flat["clean"] = True

    def _capture_failed(self, what, e):
        """The eager first execution already WAS this update, so nothing is lost either way: raise (default), or fall back
        to eager execution for the rest of the run when the caller allowed it."""
        import sys
        torch.cuda.synchronize()
        if not self.allow_eager_fallback:
            raise RuntimeError(f"hipGraph capture of the {what} failed ({e}); pass allow_eager_fallback=True (or "
                               f"use_graphs=False) to run eagerly") from e
        print(f"[gank] hipGraph capture of the {what} failed ({e}); running eagerly", file=sys.stderr)
        self.use_graphs = False

    def _agree_on_capture(self):
        """A rank whose capture of a collective failed replays graph / eager collective / graph while the others would replay
        one graph: the SAME collectives in the same order, so it would even work -- but one rank's host then paces everyone.
        All ranks take the split form as soon as one of them has to (MIN over ranks, as bench.py does for `graphs_ok`)."""
        if self.world > 1:
            flag = torch.tensor([1 if self.capture_collectives else 0], dtype=torch.int32, device=self.device)
            _dist.all_reduce(flag, op=_dist.ReduceOp.MIN, group=self.pg)
            self.capture_collectives = bool(int(flag.item()))

    def _allreduce(self, flat):
        if self.dp:
            parallel.allreduce_sum_(flat["grads"], self.pg, self.grad_wire_dtype, single_rank_too=True)

    def _run(self, key, fwd_bwd, opt, flat):
        """fwd+bwd (graph) -> [RCCL all-reduce] -> Adam (graph)."""
        # critic updates outside data parallel: the spectral norm's backward apply rides on the optimiser launch (FUSE_SN_TAIL)
        defer = opt if (FUSE_SN_TAIL and opt is self.d_opt and not self.dp and self.sn_state is not None) else None
        fwd_bwd_plain = fwd_bwd

        def fwd_bwd():
            with Fn.defer_sn_apply(defer):
                return fwd_bwd_plain()
        if not self.use_graphs:
            fwd_bwd()
            self._allreduce(flat)
            opt.apply()
            return
        if key not in self._graphs:
            # the first call of each update runs eagerly on a side stream (this IS the step: allocator
            # warm-up, lazy init), then the same code is captured (capture executes nothing)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fwd_bwd()
                self._allreduce(flat)
                opt.apply()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            try:
                # thread_local: the RCCL watchdog thread polls events concurrently under data parallel
                whole = not self.dp or self.capture_collectives
                if whole:
                    try:
                        g1 = torch.cuda.CUDAGraph()
                        with _capture(g1):       # forward + backward [+ RCCL all-reduce] + optimiser: ONE graph
                            fwd_bwd()
                            self._allreduce(flat)
                            opt.apply()
                        self._graphs[key] = (g1, None)
                    except Exception as e:  # noqa: BLE001
                        if not self.dp:
                            raise
                        import sys
                        torch.cuda.synchronize()
                        print(f"[gank] capturing the all-reduce inside the {key!r} update graph failed ({e}); "
                              f"the collective stays between two graphs", file=sys.stderr)
                        self.capture_collectives = False
                    if self.dp:
                        self._agree_on_capture()    # (entered with the same setting on every rank, so every rank calls this)
                    if key in self._graphs:
                        if not self.dp or self.capture_collectives:
                            return
                        del self._graphs[key]       # another rank fell back: every rank takes the split form
                g1 = torch.cuda.CUDAGraph()
                with _capture(g1):
                    fwd_bwd()
                g2 = torch.cuda.CUDAGraph()
                with _capture(g2):
                    opt.apply()
                self._graphs[key] = (g1, g2)
            except Exception as e:  # noqa: BLE001
                self._capture_failed(f'{key!r} update', e)
            return
        g1, g2 = self._graphs[key]
        self._ensure_clean(flat)
        self._ensure_sn_state()
        g1.replay()
        if g2 is not None:
            self._allreduce(flat)
            g2.replay()

    def _run_plain(self, key, fn):
        """Capture-and-replay of a forward-only piece (no optimiser, no exchange)."""
        if not self.use_graphs:
            fn()
            return
        if key not in self._graphs:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fn()
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    fn()
                self._graphs[key] = (g, None)
            except Exception as e:  # noqa: BLE001
                self._capture_failed(repr(key), e)
            return
        self._graphs[key][0].replay()

    # ---- public API ---------------------------------------------------------------------------------
    def d_step(self, real_u8, labels):
        """One critic update on a uint8 [B,3072] CHW-planar batch + int labels (the feed of :616-620)."""
        self.real_u8.copy_(real_u8, non_blocking=True)
        self.real_labels.copy_(labels, non_blocking=True)
        self._run('d', self._d_forward_backward, self.d_opt, self.d_flat)
        return self.d_loss

    def g_step(self):
        """One generator update (:602-603)."""
        if self.bucketed:
            self._g_step_bucketed()
        else:
            self._run('g', self._g_forward_backward, self._g_applier, self.g_flat)
        return self.g_loss

    def train_iteration(self, batches):
        """One reference iteration (:599-620): G update (skipped at iteration 0), then N_CRITIC critic
        updates, each on the next (uint8 images, labels) pair from `batches`."""
        feed = [next(batches) for _ in range(N_CRITIC)]
        if all(d.is_cuda and l.is_cuda for d, l in feed):
            # device-resident batches: the ten slot copies are ONE launch (torch's multi-tensor copy took 15 us for the 1 MB
            # of images on 15 workgroups)
            if (N_CRITIC <= 16 and all(d.dtype == self.real_all.dtype and d.is_contiguous() and l.dtype == self.labels_all.dtype and l.is_contiguous()
                                       for d, l in feed)):
                K.copy_gather2_(self.real_all, [d.view(self.batch, OUTPUT_DIM) for d, _ in feed], self.labels_all, [l for _, l in feed])
            else:
                torch._foreach_copy_(list(self.real_all.unbind(0)), [d.view(self.batch, OUTPUT_DIM) for d, _ in feed])
                torch._foreach_copy_(list(self.labels_all.unbind(0)), [l for _, l in feed])
        else:
            for i, (data, labels) in enumerate(feed):
                self.real_all[i].copy_(data, non_blocking=True)
                self.labels_all[i].copy_(labels, non_blocking=True)
        obs = self.observer
        if self.iteration > 0:
            self.g_step()
            if obs:
                obs('after_g', 0)
        if obs:
            obs('before_gen5', 0)
        self._run_plain('gen5', self._generate_for_critic)
        if obs:
            obs('after_gen5', 0)
        # the iteration counter advances inside the LAST critic update's optimiser launch (the feed-ring slot has wrapped to 0 by then)
        # where that launch is the fused one; `d_opt.bumped` says whether the update path, as last executed or captured, carries it
        self.d_opt.bump = (self.iteration_dev, self.feed_slot) if (FUSE_SN_TAIL and not self.dp and self.sn_state is not None) else None
        try:
            for i in range(N_CRITIC):      # slot i of the feed ring: the device-side slot counter walks 0..N_CRITIC-1
                if obs:
                    obs('before_d', i)
                self._run('d_pre', self._d_forward_backward_prefetched, self.d_opt, self.d_flat)
                if obs:
                    obs('after_d', i)
        finally:
            self.d_opt.bump = None
        self.iteration += 1
        if not self.d_opt.bumped:
            K.counter_add(self.iteration_dev, 1)

    @torch.no_grad()
    def dev_disc_cost(self, real_u8, labels, z=None, real_pre=None):
        """`disc_cost` of one held-out batch, forward only -- the dev-loss evaluation of the loop (:639-647: every 100
        iterations, session.run([disc_cost]) over dev_gen()).  It is the graph of the critic update (:326-381: fakes from the
        generator towers on the REAL labels, critic on concat(real, fake), hinge loss) without the optimiser, and the critic
        runs with update_collection=None as there: EVERY evaluation advances the spectral-norm `u` vectors (sn.py:48-56) --
        the reference's behaviour, reproduced, not a side effect to be avoided.  Nothing else changes: no gradient is
        accumulated, no Adam state moves.  Returns the loss as a float (one device synchronisation)."""
        set_default_store(self.store)
        b = real_u8.shape[0] if real_pre is None else real_pre.shape[0]
        lab = labels.to(self.device, torch.int32)
        fake = Generator(b, lab, noise=z, groups=N_TOWERS, rng_state=self.rng_state)
        real = K.preprocess_real(real_u8.to(self.device), self.rng_state).reshape(b, OUTPUT_DIM) if real_pre is None else real_pre
        both = torch.cat([real, fake], 0)
        both_labels = torch.cat([lab, lab], 0)
        logits, _ = Discriminator(both, both_labels, update_collection=None)
        if self.loss_type == 'WGAN':
            return float(Fn.wgan_d_loss(logits, b))
        if self.loss_type == 'Goodfellow':
            return float(Fn.gan_pointwise_loss(logits, b, 2))
        return float(Fn.hinge_d_loss(logits, b))

    def dev_loss(self, dev_batches):
        """np.mean of disc_cost over an epoch of the dev set (:641-647); dev_batches yields (uint8 [B,3072], labels)"""
        costs = [self.dev_disc_cost(torch.as_tensor(x), torch.as_tensor(np.asarray(y))) for x, y in dev_batches]
        return float(np.mean(costs)) if costs else float('nan')

    def inception_score(self, n=50000, classifier=None, splits=10, batch_size=100):
        """get_inception_score(n) of the training script (:546-555): 50 000 samples every INCEPTION_FREQUENCY iterations
        (:634-637, `maybe_inception_score`).  classifier: images [-1, 1] NHWC float32 -> logits [b, >= 1000], e.g.
        `common.inception.inception_v3.InceptionV3.from_npz(path).logits` (the frozen graph's weights are a download)."""
        return _inception_score_of(self, n, classifier, splits, batch_size)

    def maybe_inception_score(self, classifier, n=50000):
        """The loop hook (:634-637): after `train_iteration`, `self.iteration` counts finished iterations, so the reference's
        `iteration % INCEPTION_FREQUENCY == INCEPTION_FREQUENCY - 1` reads `self.iteration % INCEPTION_FREQUENCY == 0` here.
        -> (mean, std) when due, else None"""
        if self.iteration > 0 and self.iteration % INCEPTION_FREQUENCY == 0:
            return self.inception_score(n, classifier)
        return None

    @torch.no_grad()
    def sample(self, n=100, labels=None, noise=None):
        """Fixed-noise / IS sampling path (:530-555): one Generator call of n samples, batch statistics."""
        set_default_store(self.store)
        if labels is None:
            labels = K.rng_labels(n, 10, self.rng_state)
        return Generator(n, labels, noise=noise, groups=1, rng_state=self.rng_state)


def _inception_score_of(trainer, n, classifier, splits, batch_size):
    """(:543-555)  n / 100 calls of `samples_100` -- a Generator(100, ...) pass on fresh uniform labels and fresh noise, batch
    statistics of its own 100 samples --, `((x + 1) * (255.99 / 2)).astype('int32')` on the host, reshape to [-1, 32, 32, 3],
    then common.inception.inception_score.get_inception_score: pixel values back to [-1, 1], whole classifier batches, the first
    1000 logits, softmax, exp(mean KL) over `splits` splits -> (mean, std)."""
    from ..common.inception.inception_score import get_inception_score, quantize_samples
    all_samples = []
    for _ in range(int(n / 100)):
        all_samples.append(trainer.sample(100).float().cpu().numpy())
    all_samples = np.concatenate(all_samples, axis=0)
    all_samples = quantize_samples(all_samples, for_score=True).reshape((-1, 32, 32, 3))
    return get_inception_score(all_samples, splits=splits, classifier=classifier, batch_size=batch_size)


def synthetic_batches(batch_size, device, seed=0):
    """Synthetic CIFAR-10-shaped feed: uint8 [B,3072] uniform{0..255} in CHW-planar row layout and
    int32 labels uniform{0..9} (common/data/cifar10.py:9-15 format), resident in HBM."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    pool = [(torch.randint(0, 256, (batch_size, OUTPUT_DIM), generator=g, dtype=torch.uint8).to(device),
             torch.randint(0, 10, (batch_size,), generator=g, dtype=torch.int32).to(device)) for _ in range(8)]
    i = 0
    while True:
        yield pool[i % len(pool)]
        i += 1
