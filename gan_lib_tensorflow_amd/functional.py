"""torch.autograd glue over the HIP kernels (kernels.py).  Autograd is plumbing here: it only
orders the kernel launches of the backward pass.  Conventions:

* activations and their gradients are bf16 NHWC; weights and weight gradients fp32;
* a weight that carries `.main_grad` (ParamStore.flatten) gets its gradient ACCUMULATED there by the
  wgrad kernel and autograd receives None for it -- no per-parameter gradient tensors, no packing
  before the optimiser or the RCCL all-reduce;
* fan-out of an activation goes through `fork`, whose backward is the gank add kernel, so no
  framework elementwise kernel runs anywhere in the step.
"""
import torch
from torch.autograd import Function

from . import kernels as K

BF16 = K.BF16
POOL_CONV4 = True     # 3x3 conv + 2x2 mean pool as one 4x4 stride-2 conv (4 instead of 9 taps per conv output)
CPOOL_RESIDENT = True   # ... on the LDS-resident kernels where they apply (prep kind 5; kernels.cpool_res_ok)
PHASE_UPCONV = True   # NN-upsample+3x3 conv as a phase-decomposed transposed conv (4 instead of 9 taps)
RES8_CONV = True      # 3x3 convs on 8x8 images with "rfrag" operands attached: the LDS-resident kernel (gank_res8_conv3x3)
UPCONV_WGRAD_PHASE = True   # filter gradient of UpsampleConv 3x3 in its phase form (gank_upconv3x3_wgrad)
IMG16_CONV = True     # plain 3x3 convs on 16x16 images with "rfrag" operands attached: the image-resident kernel (gank_img16_conv3x3)


# Boundaries of the backward pass (data parallel: the gradient buckets of parallel.GradBuckets end here).  A network
# marks the output of each block with `boundary(x, tag)`; when a recorder is active the tensors are collected, so the
# caller can run the backward pass in segments (torch.autograd.grad from one boundary to the previous one) and start the
# all-reduce of a finished bucket while the next segment runs.
_boundary_log = None


def boundary(x, tag):
    if _boundary_log is not None and x.requires_grad:
        _boundary_log.append((tag, x))
    return x


class record_boundaries:
    def __enter__(self):
        global _boundary_log
        self._prev, _boundary_log = _boundary_log, []
        return _boundary_log

    def __exit__(self, *exc):
        global _boundary_log
        _boundary_log = self._prev
        return False


# Small same-shape filter gradients are not launched where autograd reaches them but collected and issued together
# (gank_conv2d_wgrad_batched): the critic's four 8x8x128 convs fill 144 workgroups each for a few microseconds; one
# launch of all four overlaps their latencies.  Flushed before anything reads the gradients (join_wgrad).
BATCH_SMALL_WGRADS = False     # switched on by a caller that guarantees a join_wgrad() after backward (SNGANTrainer._backward)
_deferred = {}


def _defer_wgrad(x, g, tgt, btgt, hw, k, flags):
    _deferred.setdefault((tuple(x.shape), tuple(g.shape), hw, k, flags), []).append((x, g, tgt, btgt))


_deferred_narrow = []      # filter gradients of 3-channel-input layers (the image side of a critic): issued in pairs


def _defer_narrow(x, g, tgt, btgt, hw, k):
    _deferred_narrow.append((x, g, tgt, btgt, hw, k))


def narrow_wgrad_ok(cin, cout, k, flags_free):
    """a filter gradient the streaming 3-channel-input kernel takes (gank_conv2d_wgrad_narrow_pair)"""
    return BATCH_SMALL_WGRADS and flags_free and cin == 3 and k in (1, 3) and cout % 128 == 0


# Split-K partial results written as slabs instead of added with fp32 atomics (in a critic update the atomics of the fused
# image-side gradient and of the batched 8x8 layers were 17 + 12 us of their kernels' 40 + 30): the jobs that sum them collect
# here and leave as ONE launch in join_wgrad().  Only under BATCH_SMALL_WGRADS (the caller guarantees the join).
SLAB_WGRADS = True
_slab_jobs = []
# ... and the label-gradient launch of a critic update (concat_label_conv1's backward) rides on that launch as extra workgroups; what
# depends on it (the label branch's dense-layer gradients) is issued right behind: entries (label arguments of kernels.sum_slabs, finish(parts))
LABEL_BWD_IN_SUM_SLABS = True
_label_jobs = []


def flush_wgrads():
    for (_, _, hw, k, flags), items in _deferred.items():
        K.conv2d_wgrad_batched(items, hw, k, flags, 1.0, slab_jobs=_slab_jobs if SLAB_WGRADS else None)
    _deferred.clear()
    if _slab_jobs:
        if _label_jobs and sum(c for _, c, _ in _slab_jobs) <= K.SUM_SLABS_MAX_JOBS:
            label, finish = _label_jobs.pop(0)
            finish(K.sum_slabs(_slab_jobs, label))
        else:
            K.sum_slabs(_slab_jobs)
    for label, finish in _label_jobs:          # (no summing launch to ride on: a launch of their own)
        finish(K.label_conv3x3_bwd_pooled(*label))
    _label_jobs.clear()
    while len(_deferred_narrow) >= 2:
        a, b = _deferred_narrow.pop(0), _deferred_narrow.pop(0)
        K.conv2d_wgrad_narrow_pair(a, b)
    for x, g, tgt, btgt, hw, k in _deferred_narrow:
        K.conv2d_wgrad(x, g, tgt, hw, k, 0, 1.0, dbias=btgt)
    _deferred_narrow.clear()


def reset_deferred():
    """Drop every deferred filter gradient (start of a backward pass, and after one that raised or whose
    capture aborted): stale (x, dy) pairs must never be flushed into another pass's gradient buffers."""
    _deferred.clear()
    _deferred_narrow.clear()
    _slab_jobs.clear()
    _label_jobs.clear()


def join_wgrad():
    """Every filter gradient issued or deferred so far is complete / in stream order (before the optimiser and the
    SN backward)."""
    if _deferred or _deferred_narrow or _slab_jobs or _label_jobs:
        flush_wgrads()


def _target(p):
    """Where a weight gradient is written: (buffer, accumulated_in_place).  `main_grad` = view of the
    network's flat gradient buffer (autograd gets None); `_grad_buf` = pre-zeroed view handed out by a
    batched producer (spectral norm) and returned to autograd as the gradient; else a fresh zero tensor."""
    mg = getattr(p, "main_grad", None)
    if mg is not None:
        return mg, True
    gb = getattr(p, "_grad_buf", None)
    if gb is not None:
        return gb, False
    return (K.zeros_f32(p.shape, p.device) if p.dtype == torch.float32 and p.is_cuda else torch.zeros_like(p)), False


def _prepared(W, k, cin, cout, want_f, want_d):
    """bf16 MFMA operand layouts of W: taken from `W._prep` when a batched preparation attached them."""
    prep = getattr(W, "_prep", None)
    if prep is not None and (not want_f or prep[0] is not None) and (not want_d or prep[1] is not None):
        return prep
    return K.prep_weights(W.detach().view(k, k, cin, cout), want_f, want_d)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


CONV1X1_BWD_ONE_LAUNCH = True   # a 1x1 conv's input gradient as extra workgroups of its filter-gradient launch (gank_conv1x1_wgrad_dgrad)
FUSE_IMAGE_WGRAD = True   # D.Block.1.Conv1's and D.Block.1.Shortcut's filter gradients inside the ConvMeanPool input-gradient launch (round 5)


class ImageWgradSink:
    """Links a 3-channel-input conv whose input needs NO gradient (conv_1 / the 1x1 shortcut of OptimizedResBlockDisc1 in a critic
    update, gan_cifar_resnet.py:212-234) with the resident ConvMeanPool behind it.  The ConvMeanPool's input gradient then has
    exactly one consumer -- this conv's filter and bias gradient -- and gank_cpool_res_dgrad_image_wgrad computes both inside
    the launch that would have produced the 33.5-MB tensor (never stored).  The producer's forward creates the sink (on its
    output tensor: `_image_wgrad_sink` / `_shortcut_sink`), the ConvMeanPool's forward picks it up, its backward fills `tgt`
    and sets `done`, and the producer's backward -- called with no gradient -- hands `tgt` to autograd."""
    __slots__ = ("x", "W", "bias", "done", "tgt")

    def __init__(self, x, W, bias):
        self.x, self.W, self.bias, self.done, self.tgt = x, W, bias, False, None

    def targets(self):
        """(filter target, accumulated?, bias target | None, accumulated?) -- resolved once, by whoever runs the fused launch"""
        tgt, acc = _target(self.W)
        btgt, bacc = (None, True)
        if self.bias is not None and self.bias.requires_grad:
            btgt, bacc = _target(self.bias)
        self.tgt = (tgt, acc, btgt, bacc)
        return self.tgt


class ShortcutLink:
    """Identity-shortcut residual block (`shortcut + conv_2(relu(conv_1(relu(x))))`, no normalisation): the gradient of
    the block input is mask(dgrad_1) + dy.  conv_2's backward parks dy here instead of returning it for the shortcut
    tensor, and conv_1's input-gradient kernel adds it in its epilogue: one add launch less per block and backward pass
    (the critic's 8x8 blocks: 12 launches of ~6 us per iteration)."""
    __slots__ = ("armed", "g")

    def __init__(self):
        self.armed = False
        self.g = None


class _Conv2d(Function):
    """conv2d SAME stride 1 with fused NN-upsample / relu on the input and bias / mean-pool /
    residual / tanh on the output (common/ops/conv2d.py:180-216; gan_cifar_resnet.py:112-153)."""

    last_stats = None
    last_sink = None

    @staticmethod
    def forward(ctx, x, W, bias, residual, upsample, in_relu, pool_out, out_tanh, stats_groups=0):
        _Conv2d.last_stats = None
        _Conv2d.last_sink = None
        ctx.sink = ctx.isink = ctx.ssink = None
        if W.dim() == 2:
            k, cin, cout = 1, W.shape[0], W.shape[1]
        else:
            k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
        assert x.dim() == 4 and x.shape[3] == cin, (tuple(x.shape), tuple(W.shape))
        assert not (pool_out and (out_tanh or upsample))
        n, h, w, _ = x.shape
        H, Wd = (2 * h, 2 * w) if upsample else (h, w)
        flags = (K.IN_UPSAMPLE2X if upsample else 0) | (K.IN_RELU if in_relu else 0) | (K.OUT_TANH if out_tanh else 0)
        b = bias.detach() if bias is not None else None
        # a shortcut computed at half resolution (1x1 conv commuted with the upsample): the epilogue adds it
        # nearest-neighbour upsampled; paths without that epilogue get it materialised
        res_up = residual is not None and getattr(residual, "_up2x", False)
        ctx.res_up_orig = res_up
        # 8x8 images with the "rfrag" operands attached: one LDS-resident image per workgroup (conv_resident.hip), the
        # NN-upsample of a 4x4 input done by its loader
        res8 = (RES8_CONV and k == 3 and not in_relu and not pool_out and not out_tanh and getattr(W, "_prep_res", None) is not None
                and K.res8_conv3x3_ok(n, (H, Wd), cin, cout))
        # plain 3x3 on 16x16 images with the "rfrag" operands attached: one image x 128 output channels per workgroup
        img16 = (IMG16_CONV and k == 3 and not upsample and not pool_out and not out_tanh
                 and getattr(W, "_prep_res", None) is not None and K.img16_conv3x3_ok(n, (H, Wd), cin, cout))
        # NN-upsample + 3x3: run as the 4 output phases of the equivalent 4x4 stride-2 transposed conv
        phase = upsample and k == 3 and cin % 64 == 0 and not in_relu and PHASE_UPCONV and not res8
        # 3x3 conv + 2x2 mean pool: run as ONE 4x4 stride-2 conv (16 taps per pooled pixel = 4 per conv output)
        pool4 = pool_out and k == 3 and cin % 64 == 0 and cout % 64 == 0 and POOL_CONV4
        if res_up and (phase or pool_out):
            residual = K.unpool2x2_add(residual, None, 1.0)
            res_up = False
        if res8:
            rflags = (K.IN_UPSAMPLE2X if upsample else 0) | (K.RES_UPSAMPLE2X if res_up else 0)
            if stats_groups:
                y, _Conv2d.last_stats = K.res8_conv3x3(x, W._prep_res[0], b, cout, rflags, residual, stats_groups)
            else:
                y = K.res8_conv3x3(x, W._prep_res[0], b, cout, rflags, residual)
        elif img16:
            iflags = (K.IN_RELU if in_relu else 0) | (K.RES_UPSAMPLE2X if res_up else 0)
            if stats_groups:
                y, _Conv2d.last_stats = K.img16_conv3x3(x, W._prep_res[0], b, cout, iflags, None, residual, stats_groups)
            else:
                y = K.img16_conv3x3(x, W._prep_res[0], b, cout, iflags, None, residual)
        elif phase:
            wph, _ = getattr(W, "_prep_up", None) or K.upconv3x3_prep(W.detach().view(3, 3, cin, cout))
            if stats_groups and not out_tanh:
                y, _Conv2d.last_stats = K.upconv3x3_fprop(x, wph, b, cout, 0, residual, stats_groups)
            else:
                y = K.upconv3x3_fprop(x, wph, b, cout, K.OUT_TANH if out_tanh else 0, residual)
        elif pool4 and getattr(W, "_prep_cpres", None) is not None:       # resident form (conv_resident.hip)
            y = K.cpool_res_fprop(x, W._prep_cpres[0], b, cout, K.IN_RELU if in_relu else 0, residual)
        elif pool4:
            wp4, _ = getattr(W, "_prep_pool", None) or K.convpool3x3_prep(W.detach().view(3, 3, cin, cout))
            y = K.convpool3x3_fprop(x, wp4, b, cout, K.IN_RELU if in_relu else 0, residual)
        elif pool_out:
            wf, _ = _prepared(W, k, cin, cout, True, False)
            yfull = K.conv2d_fprop(x, wf, b, (H, Wd), cout, k, flags)
            y = K.pool2x2(yfull, 0.25, residual)
        else:
            pre = getattr(x, "_conv_result", None)
            if (pre is not None and pre[0] is W and pre[1] is bias and flags == 0 and residual is None and not stats_groups
                    and tuple(pre[2].shape) == (n, H, Wd, cout)):
                y = pre[2]          # computed by the launch that produced x (concat_label_conv1: the 1x1 shortcut on the pooled concatenation)
            else:
                wf, _ = _prepared(W, k, cin, cout, True, False)
                if stats_groups and not out_tanh:
                    y, _Conv2d.last_stats = K.conv2d_fprop(x, wf, b, (H, Wd), cout, k, flags | (K.RES_UPSAMPLE2X if res_up else 0), 1.0, residual,
                                                           stats_groups=stats_groups)
                else:
                    y = K.conv2d_fprop(x, wf, b, (H, Wd), cout, k, flags | (K.RES_UPSAMPLE2X if res_up else 0), 1.0, residual)
        ctx.res_up = res_up
        # identity-shortcut fusion (ShortcutLink): conv_1 arms the link when it will produce a plain input gradient;
        # conv_2 (called after it) then parks dy for it instead of returning it along the shortcut
        ctx.add_link = ctx.res_link = None
        link = getattr(x, "_add_link", None)
        if link is not None and ctx.needs_input_grad[0] and not (upsample or pool_out or phase or pool4 or res8):
            link.armed = True
            ctx.add_link = link
        rlink = getattr(residual, "_grad_link", None) if residual is not None else None
        if rlink is not None and rlink.armed and not res_up and not ctx.res_up_orig:
            ctx.res_link = rlink
        ctx.save_for_backward(x, W, y if out_tanh else None)
        ctx.cfg = (k, cin, cout, H, Wd, upsample, in_relu, pool_out, out_tanh, bias, phase, pool4)
        if FUSE_IMAGE_WGRAD and x.is_cuda:
            if (k == 3 and cin == 3 and residual is None and not (upsample or in_relu or pool_out or out_tanh or stats_groups)
                    and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]):
                # the image-side conv of a critic update: a ConvMeanPool behind it may take over the filter gradient (ImageWgradSink)
                ctx.sink = _Conv2d.last_sink = ImageWgradSink(x, W, bias)
                ctx.set_materialize_grads(False)           # ... and then calls this node's backward with no gradient at all
            elif (pool4 and in_relu and getattr(W, "_prep_cpres", None) is not None and ctx.needs_input_grad[0]
                  and getattr(x, "_image_wgrad_sink", None) is not None and K.cpool_res_dgrad_image_wgrad_ok(y, cin)):
                ctx.isink = x._image_wgrad_sink
                ctx.ssink = getattr(residual, "_shortcut_sink", None) if (residual is not None and not ctx.res_up_orig) else None
        # the backward pass runs the same kernel with the channel roles swapped: its own geometry check (Cin there = cout here),
        # otherwise the generic input-gradient path
        ctx.res8 = res8 and W._prep_res[1] is not None and K.res8_conv3x3_ok(n, (8, 8), cout, cin)
        ctx.img16 = img16 and W._prep_res[1] is not None and K.img16_conv3x3_ok(n, (16, 16), cout, cin)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        k, cin, cout, H, Wd, upsample, in_relu, pool_out, out_tanh, bias, phase, pool4 = ctx.cfg
        if ctx.sink is not None and ctx.sink.done:        # the ConvMeanPool behind this conv computed its filter / bias gradient
            tgt, acc, btgt, bacc = ctx.sink.tgt
            return None, (None if acc else tgt), (None if (btgt is None or bacc) else btgt), None, None, None, None, None, None
        if dy is None:
            return (None,) * 9
        g = _c(dy)
        if out_tanh:
            g = K.tanh_bwd(g, y)
        scale = 0.25 if pool_out else 1.0
        dW = db = dx = None
        dx_fused = None
        btgt = None
        if bias is not None and ctx.needs_input_grad[2]:
            btgt, bacc = _target(bias)
            db = None if bacc else btgt
        if ctx.needs_input_grad[1] and pool4:
            tgt, acc = _target(W)
            K.convpool3x3_wgrad(x, g, tgt.view(3, 3, cin, cout), K.IN_RELU if in_relu else 0, dbias=btgt,
                                slab_jobs=_slab_jobs if (SLAB_WGRADS and BATCH_SMALL_WGRADS) else None)
            dW = None if acc else tgt
        elif ctx.needs_input_grad[1]:
            tgt, acc = _target(W)
            wflags = (K.IN_UPSAMPLE2X if upsample else 0) | (K.IN_RELU if in_relu else 0) | (K.DY_UPSAMPLE2X if pool_out else 0)
            small = (BATCH_SMALL_WGRADS and not upsample and not pool_out and k == 3 and cin % 128 == 0 and cout % 128 == 0
                     and x.shape[0] * H * Wd <= 8192)
            if small:
                _defer_wgrad(x, g, tgt, btgt, (H, Wd), k, wflags)
            elif narrow_wgrad_ok(cin, cout, k, wflags == 0):
                _defer_narrow(x, g, tgt, btgt, (H, Wd), k)       # paired with the block's other image-side layer (one launch)
            elif UPCONV_WGRAD_PHASE and upsample and k == 3 and not in_relu and not pool_out and K.upconv3x3_wgrad_ws(x, cout) > 0:
                # phase form: 16 (phase, tap) products at LOW resolution instead of 9 taps on the upsampled input (4/9 of the work)
                K.upconv3x3_wgrad(x, g, tgt.view(3, 3, cin, cout))
                if btgt is not None:
                    K.colsum(g, btgt, 1.0)           # (the rows kernel sums its OTHER operand: the bias gradient is a launch of its own)
            elif (CONV1X1_BWD_ONE_LAUNCH and k == 1 and wflags == 0 and scale == 1.0 and ctx.needs_input_grad[0] and ctx.add_link is None
                  and not ctx.img16 and not ctx.res8 and not phase
                  and int(K.lib().gank_conv2d_wgrad_slab_elems(x.shape[0], H, Wd, cin, cout, 1, 0)) == 0
                  and K.conv1x1_wgrad_dgrad_ok(x.shape[0], (H, Wd), cin, cout, _prepared(W, k, cin, cout, False, True)[1])):
                # a 1x1 layer's two gradients in one launch: the input gradient from extra workgroups of the filter-gradient launch
                dx_fused = K.conv1x1_wgrad_dgrad(x, g, tgt, _prepared(W, k, cin, cout, False, True)[1], dbias=btgt)
            else:
                # bias gradient rides on the dy stream
                K.conv2d_wgrad(x, g, tgt, (H, Wd), k, wflags, scale, dbias=btgt,
                               slab_jobs=_slab_jobs if (SLAB_WGRADS and BATCH_SMALL_WGRADS) else None)
            dW = None if acc else tgt
        elif btgt is not None:
            K.colsum(g, btgt, 1.0)
        if dx_fused is not None:
            dx = dx_fused
        elif ctx.needs_input_grad[0] and ctx.res8:
            # the conv with the dgrad operand (taps flipped, channels swapped); behind an upsample its 2x2 sums
            dx = K.res8_conv3x3(g, W._prep_res[1], None, cin, K.OUT_POOLSUM2X if upsample else 0)
        elif ctx.needs_input_grad[0] and phase:
            prep = getattr(W, "_prep_up", None) or K.upconv3x3_prep(W.detach().view(3, 3, cin, cout))
            dx = K.upconv3x3_dgrad(g, prep[1], cin)       # 4x4 stride-2 conv of dy: no hi-res dgrad, no 2x2 sum
        elif ctx.needs_input_grad[0] and pool4 and getattr(W, "_prep_cpres", None) is not None and ctx.isink is not None:
            # the input gradient's only consumer is the image-side conv's filter gradient: both in one launch, nothing stored
            isink, ssink = ctx.isink, ctx.ssink
            t1, _, b1, _ = isink.targets()
            ts = bs_ = xp = None
            if ssink is not None:
                ts, _, bs_, _ = ssink.targets()
                xp = ssink.x
            K.cpool_res_dgrad_image_wgrad(g, W._prep_cpres[1], x, isink.x, t1, b1, xp, ts, bs_,
                                          slab_jobs=_slab_jobs if (SLAB_WGRADS and BATCH_SMALL_WGRADS) else None)
            isink.done = True
            if ssink is not None:
                ssink.done = True
            dx = None
        elif ctx.needs_input_grad[0] and pool4 and getattr(W, "_prep_cpres", None) is not None:
            dx = K.cpool_res_dgrad(g, W._prep_cpres[1], cin, x if in_relu else None)
        elif ctx.needs_input_grad[0] and pool4:
            prep = getattr(W, "_prep_pool", None) or K.convpool3x3_prep(W.detach().view(3, 3, cin, cout))
            dx = K.convpool3x3_dgrad(g, prep[1], cin, x if in_relu else None)
        elif ctx.needs_input_grad[0]:
            wd = None if ctx.img16 else _prepared(W, k, cin, cout, False, True)[1]
            dflags = K.IN_UPSAMPLE2X if pool_out else 0
            if upsample:
                dxf = K.conv2d_dgrad(g, wd, (H, Wd), cin, k, dflags, scale)
                dx = K.pool2x2(dxf, 1.0)                      # gradient of the NN-upsample: 2x2 sum
                if in_relu:
                    dx = K.relu_bwd(dx, x)
            else:
                extra = None
                if ctx.add_link is not None:
                    extra, ctx.add_link.g = ctx.add_link.g, None
                if ctx.img16:      # the same kernel with the dgrad operand (taps flipped, channels swapped); relu mask and fan-in in its epilogue
                    dx = K.img16_conv3x3(g, W._prep_res[1], None, cin, 0, x if in_relu else None, extra)
                else:
                    dx = K.conv2d_dgrad(g, wd, (H, Wd), cin, k, dflags, scale, extra, x if in_relu else None)
        dres = None
        if ctx.needs_input_grad[3] and ctx.res_link is not None:
            ctx.res_link.g = g                     # consumed by conv_1's input-gradient epilogue
        elif ctx.needs_input_grad[3]:
            # gradient of the (upsampled) shortcut add: dy itself, or its 2x2 sums for a half-resolution shortcut
            dres = K.pool2x2(g, 1.0) if (ctx.res_up or getattr(ctx, "res_up_orig", False)) else g
        return dx, dW, db, dres, None, None, None, None, None


FEW_OUT_UPCONV = True    # upsample + conv with <= 4 output channels (Pix2Pix decoder_1): 1x1 conv at low resolution + tap gather (gank_tap_gather_up2)


class _ConvGeneral(Function):
    """Convolution with any filter size (even ones too), stride 1 or 2 and an explicit leading pad -- the 4x4 convs of the
    Pix2Pix U-Net and PatchGAN critic (Pix2Pix/networks.py:366-536).  `pad` rows / columns of zeros in front; the output
    size says how far the window runs past the other edge (TF SAME and tf.pad + VALID are both this)."""

    @staticmethod
    def forward(ctx, x, W, bias, stride, pad, out_hw, upsample, in_relu, out_tanh):
        k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
        assert x.dim() == 4 and x.shape[3] == cin, (tuple(x.shape), tuple(W.shape))
        ctx.few = (FEW_OUT_UPCONV and upsample and stride == 1 and cout <= 4 and k * k * cout <= 64 and cin % 64 == 0
                   and tuple(out_hw) == (2 * x.shape[1], 2 * x.shape[2]))
        if ctx.few:
            # <= 4 output channels behind the upsample: a 1x1 conv at low resolution to the k*k*Cout tap partials + a tap gather
            wz = K.fewout_pack(W.detach().view(k, k, cin, cout), 64)          # [Cin, 64]: column t * Cout + co, zeros behind
            wf, wd = K.prep_weights(wz.view(1, 1, cin, 64), True, True)
            Z = K.conv2d_fprop(x, wf, None, (x.shape[1], x.shape[2]), 64, 1, K.IN_RELU if in_relu else 0)
            y = K.tap_gather_up2(Z, bias.detach() if bias is not None else None, k, pad, cout, out_tanh)
            ctx.wd_few = wd
            ctx.save_for_backward(x, W, y if out_tanh else None)
            ctx.cfg = (k, cin, cout, stride, pad, upsample, in_relu, out_tanh, bias)
            return y
        wf, _ = _prepared(W, k, cin, cout, True, False)
        flags = (K.IN_UPSAMPLE2X if upsample else 0) | (K.IN_RELU if in_relu else 0) | (K.OUT_TANH if out_tanh else 0)
        y = K.conv2d_general_fprop(x, wf, bias.detach() if bias is not None else None, out_hw, cout, k, stride, pad, flags)
        ctx.save_for_backward(x, W, y if out_tanh else None)
        ctx.cfg = (k, cin, cout, stride, pad, upsample, in_relu, out_tanh, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        k, cin, cout, stride, pad, upsample, in_relu, out_tanh, bias = ctx.cfg
        g = _c(dy)
        if out_tanh:
            g = K.tanh_bwd(g, y)
        dW = db = dx = None
        btgt = None
        if bias is not None and ctx.needs_input_grad[2]:
            btgt, bacc = _target(bias)
            db = None if bacc else btgt
        if ctx.few:
            n, h, w, _ = x.shape
            col = K.tap_scatter_up2(g, k, pad, 64)                  # gradient of the tap partials Z
            if ctx.needs_input_grad[1]:
                tgt, acc = _target(W)
                tmp = K.zeros_f32((1, 1, cin, 64), x.device)
                K.conv2d_wgrad(x, col, tmp, (h, w), 1, K.IN_RELU if in_relu else 0, 1.0)
                K.fewout_pack_bwd(tmp.view(cin, 64), tgt.view(k, k, cin, cout))
                dW = None if acc else tgt
            if btgt is not None:
                K.colsum(g, btgt, 1.0)
            if ctx.needs_input_grad[0]:
                dx = K.conv2d_dgrad(col, ctx.wd_few, (h, w), cin, 1, 0, 1.0, None, x if in_relu else None)
            return dx, dW, db, None, None, None, None, None, None
        if ctx.needs_input_grad[1]:
            tgt, acc = _target(W)
            K.conv2d_general_wgrad(x, g, tgt, k, stride, pad, (K.IN_UPSAMPLE2X if upsample else 0) | (K.IN_RELU if in_relu else 0), dbias=btgt)
            dW = None if acc else tgt
        elif btgt is not None:
            K.colsum(g, btgt, 1.0)
        if ctx.needs_input_grad[0]:
            n, h, w, _ = x.shape
            if stride == 1:
                _, wd = _prepared(W, k, cin, cout, False, True)
                if upsample:
                    dx = K.pool2x2(K.conv2d_general_dgrad(g, wd, (2 * h, 2 * w), cin, k, pad), 1.0)       # gradient of the NN-upsample: 2x2 sum
                    if in_relu:
                        dx = K.relu_bwd(dx, x)
                else:
                    dx = K.conv2d_general_dgrad(g, wd, (h, w), cin, k, pad, x if in_relu else None)
            else:
                # transposed conv by output phase: the filter memory [k,k,Cin,Cout] IS the transposed conv's filter [k,k,Cout',Cin']
                if not (k in (3, 4) and cout % 64 == 0 and pad == (max(k - 2, 0)) // 2):
                    raise NotImplementedError(f"stride-2 input gradient needs a 3x3 / 4x4 filter, its SAME pad and Cout % 64 == 0 (k={k}, pad={pad}, Cout={cout})")
                dx = K.upconv3x3_fprop(g, K.deconv2d_prep_phases(W.detach().view(k, k, cin, cout)), None, cin)
                if dx.shape[1] != h or dx.shape[2] != w:
                    dx = dx[:, :h, :w, :].contiguous()            # an odd input size (the 1x1 bottom of a U-Net)
                if in_relu:
                    dx = K.relu_bwd(dx, x)
        return dx, dW, db, None, None, None, None, None, None


# NN-upsample + 4x4 SAME conv (Pix2Pix decoders) by output phase: output row 2i + a reads the low-resolution rows
#   a = 0: i-1 (ky 0), i (ky 1 + ky 2), i+1 (ky 3);     a = 1: i (ky 0 + ky 1), i+1 (ky 2 + ky 3)
# so all four phases are 3x3 convs of the low-resolution input whose filters are sums of the 4x4 filter's taps.
PHASE_STACK_MIN_PIXELS = 16384    # low-resolution pixels; below, the stacked filter (9/4 of the 4x4 one, rebuilt and folded back every pass) costs more than the taps it saves
PHASE_STACK_UPCONV4 = True    # ... run as ONE 3x3 conv to 4 Cout channels (phase-major) + depth_to_space: 36 instead of 64 taps per 2x2 outputs, on the patch / two-group kernels


class _PhaseStack4(Function):
    """[4,4,Cin,Cout] filter -> the stacked 3x3 filter [3,3,Cin,4 Cout] of the four output phases (phase-major output channels);
    backward folds the stacked gradient back and adds it to wherever the filter's gradient lives"""

    @staticmethod
    def forward(ctx, W):
        ctx.W = W
        return K.phase_stack4(W.detach().contiguous())

    @staticmethod
    def backward(ctx, g3):
        W = ctx.W
        tgt, acc = _target(W)
        K.phase_stack4_bwd(_c(g3), tgt.view(W.shape))
        return None if acc else tgt


class _Tile4(Function):
    """bias [C] -> [4 C] (one copy per output phase); backward: the four gradients summed into the bias gradient"""

    @staticmethod
    def forward(ctx, b):
        ctx.b = b
        return K.tile_rows(b.detach(), 4)

    @staticmethod
    def backward(ctx, g):
        tgt, acc = _target(ctx.b)
        K.tile_rows_bwd(_c(g), tgt.view(-1), 4)
        return None if acc else tgt


class _DepthToSpace2(Function):
    @staticmethod
    def forward(ctx, x):
        return K.depth_to_space2(_c(x))

    @staticmethod
    def backward(ctx, g):
        return K.space_to_depth2(_c(g))


def conv2d_general(x, W, bias=None, stride=1, pad=0, out_hw=None, upsample=False, in_relu=False, out_tanh=False):
    k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
    if (PHASE_STACK_UPCONV4 and upsample and k == 4 and stride == 1 and pad == 1 and not out_tanh and cin % 64 == 0 and cout % 32 == 0
            and x.shape[0] * x.shape[1] * x.shape[2] >= PHASE_STACK_MIN_PIXELS and x.shape[1] % 8 == 0 and x.shape[2] % 16 == 0 and tuple(out_hw) == (2 * x.shape[1], 2 * x.shape[2])):
        y3 = _Conv2d.apply(x, _PhaseStack4.apply(W), _Tile4.apply(bias) if bias is not None else None, None, False, in_relu, False, False, 0)
        return _DepthToSpace2.apply(y3)
    return _ConvGeneral.apply(x, W, bias, int(stride), int(pad), (int(out_hw[0]), int(out_hw[1])), upsample, in_relu, out_tanh)


import os as _os
CONV_EPILOGUE_STATS = _os.environ.get("GANK_EPILOGUE_STATS", "1") == "1"     # batch-norm statistics of a conv's output from its own epilogue (two-group kernel), where asked for


class _PadChannels(Function):
    """x [N,H,W,C] -> [N,H,W,Cp] with zero channels behind (one concat launch; backward: the first C channels of the gradient)"""

    @staticmethod
    def forward(ctx, x, cp):
        ctx.c = x.shape[3]
        return K.pad_rows(_c(x), cp)

    @staticmethod
    def backward(ctx, g):
        return K.split_channels(_c(g), ctx.c)[0], None


class _PadCin(Function):
    """filter [k,k,Cin,Cout] -> [k,k,Cp,Cout] with zero input channels behind; backward: the gradient of the real channels, added
    to wherever the filter's gradient lives (`_target`)"""

    @staticmethod
    def forward(ctx, W, cp):
        ctx.W = W
        k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
        return K.pad_rows(W.detach().contiguous().view(k * k, cin * cout), cp * cout).view(k, k, cp, cout)

    @staticmethod
    def backward(ctx, g):
        W = ctx.W
        tgt, acc = _target(W)
        k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
        K.pad_rows_bwd(_c(g).view(k * k, g.shape[2] * cout), tgt.view(k * k, cin * cout))
        return (None if acc else tgt), None


PAD_ODD_CIN = True    # a plain conv whose Cin > 64 is not a multiple of 64 (PGGAN: 513 channels behind minibatch-std, model_nvidia.py:128-129)
#                       runs on zero-padded operands (576 channels): the MFMA kernels instead of the K-packed scalar gather


def conv2d(x, W, bias=None, residual=None, upsample=False, in_relu=False, pool_out=False, out_tanh=False, stats_groups=0):
    """stats_groups > 0: the output feeds a (conditional) batch norm over that many towers; when the kernel that runs can
    accumulate its statistics, they ride along on the result (`y._cbn_stats`) and cond_batchnorm skips its statistics pass."""
    if PAD_ODD_CIN and W.dim() == 4 and x.shape[3] > 64 and x.shape[3] % 64 and not upsample and not pool_out and x.is_cuda:
        cp = (x.shape[3] + 63) // 64 * 64
        x, W = _PadChannels.apply(x, cp), _PadCin.apply(W, cp)
    y = _Conv2d.apply(x, W, bias, residual, upsample, in_relu, pool_out, out_tanh, stats_groups if CONV_EPILOGUE_STATS else 0)
    if stats_groups and _Conv2d.last_stats is not None:
        y._cbn_stats = _Conv2d.last_stats
        _Conv2d.last_stats = None
    if _Conv2d.last_sink is not None:
        y._image_wgrad_sink, _Conv2d.last_sink = _Conv2d.last_sink, None
    return y


def _res_chain8_grads(ctx, g, head=None):
    """the backward launch of a fused chain and its filter / bias gradients -> (dx, [grads of the 4 * nb parameters])"""
    nb, pool, Ws, bs = ctx.cfg
    x, h1s, ys = ctx.kept
    off = ctx.param_offset
    need_w = [ctx.needs_input_grad[off + 4 * b + 2 * j] for b in range(nb) for j in range(2)]
    need_b = [bs[i] is not None and ctx.needs_input_grad[off + 2 * i + 1] for i in range(2 * nb)]
    train = any(need_w) or any(need_b)
    xins = [x] + ys[:-1]
    if head is not None:
        dx, g1s, dys = K.res8_chain_bwd(None, None, ys[-1], [w._prep_res[1] for w in Ws], h1s, xins, keep=train, head=head)
    else:
        g = _c(g)
        dx, g1s, dys = K.res8_chain_bwd(None if pool else g, g if pool else None, ys[-1] if pool else None,
                                        [w._prep_res[1] for w in Ws], h1s, xins, keep=train)
    grads = [None] * (4 * nb)
    if train:
        for b in range(nb):
            for j, (xop, gop) in enumerate(((xins[b], g1s[b]), (h1s[b], dys[b]))):      # conv_1: (relu(x), g1); conv_2: (relu(h1), dy)
                i = 2 * b + j
                W, bias = Ws[i], bs[i]
                btgt = None
                if need_b[i]:
                    btgt, bacc = _target(bias)
                    grads[4 * b + 2 * j + 1] = None if bacc else btgt
                if need_w[i]:
                    tgt, acc = _target(W)
                    grads[4 * b + 2 * j] = None if acc else tgt
                    if BATCH_SMALL_WGRADS:
                        _defer_wgrad(xop, gop, tgt, btgt, (8, 8), 3, K.IN_RELU)
                    else:
                        K.conv2d_wgrad(xop, gop, tgt, (8, 8), 3, K.IN_RELU, 1.0, dbias=btgt)
                elif btgt is not None:
                    K.colsum(gop, btgt, 1.0)
    return dx, grads


class _ResChain8(Function):
    """Up to two identity-shortcut residual blocks on 8x8x128 images in one launch each way (conv_resident.hip):
    y = x + conv_2(relu(conv_1(relu(x)) + b1)) + b2 per block (gan_cifar_resnet.py:176-209 with resample=None and no
    normalisation), optionally followed by relu + spatial mean (:299-301).  params = (W1, b1, W2, b2) per block; every W
    carries `_prep_res` (prep kind 4)."""

    @staticmethod
    def forward(ctx, x, pool, grad_on, *params):
        nb = len(params) // 4
        Ws = [params[4 * b + 2 * j] for b in range(nb) for j in range(2)]
        bs = [params[4 * b + 2 * j + 1] for b in range(nb) for j in range(2)]
        keep = grad_on and any(ctx.needs_input_grad)      # (needs_input_grad ignores torch.no_grad(); grad mode is off inside forward)
        out, h1s, ys = K.res8_chain_fwd(x, [w._prep_res[0] for w in Ws], [b.detach() if b is not None else None for b in bs], keep, pool)
        ctx.cfg = (nb, pool, Ws, bs)
        ctx.param_offset = 3
        ctx.kept = (x, h1s, ys) if keep else None
        return out

    @staticmethod
    def backward(ctx, g):
        dx, grads = _res_chain8_grads(ctx, g)
        return (dx if ctx.needs_input_grad[0] else None, None, None, *grads)


class HingeHeadSpec:
    """The critic's last dense layer + hinge loss as a `loss_head` of Discriminator(): mode 0 = hinge_d with the first n_real
    rows real (gan_cifar_resnet.py:379-381), mode 1 = hinge_g (:492); `out`: persistent fp32[1] buffer that receives the loss;
    loss_scale: see grad_seed.  Called with (features, W, b) it is hinge_d_head / hinge_g_head (one launch of its own); a model
    that ends in the fused 8x8 chain hands it to res_chain8 instead, which computes the logits in the chain's forward launch
    and the loss, its derivative and the layer's gradients in the chain's backward launch (no launch of its own)."""

    def __init__(self, mode, n_real=0, out=None, loss_scale=1.0):
        self.mode, self.n_real, self.out, self.loss_scale = int(mode), int(n_real), out, float(loss_scale)

    def __call__(self, x, W, bias):
        if self.mode == 0:
            return hinge_d_head(x, W, bias, self.n_real, out=self.out, loss_scale=self.loss_scale)
        return hinge_g_head(x, W, bias, out=self.out, loss_scale=self.loss_scale)


class _ResChain8Head(Function):
    """_ResChain8 with pool=True and the head of HingeHeadSpec inside: forward launch -> pooled features AND logits; backward
    launch -> hinge derivative from the logits, the loss value, D.Output's weight / bias gradients, then the chain's input
    gradients as before.  The returned loss tensor holds its value once the BACKWARD launch has run (a train step always
    runs it); a call that needs no gradient evaluates the loss with the stand-alone head kernel on the pooled features."""

    @staticmethod
    def forward(ctx, x, head_W, head_b, spec, grad_on, *params):
        nb = len(params) // 4
        Ws = [params[4 * b + 2 * j] for b in range(nb) for j in range(2)]
        bs = [params[4 * b + 2 * j + 1] for b in range(nb) for j in range(2)]
        keep = grad_on and any(ctx.needs_input_grad)      # (needs_input_grad ignores torch.no_grad(); grad mode is off inside forward)
        hw = head_W.detach().reshape(-1)
        hb = head_b.detach() if head_b is not None else None
        pooled, h1s, ys, logits = K.res8_chain_fwd(x, [w._prep_res[0] for w in Ws], [b.detach() if b is not None else None for b in bs],
                                                  keep, True, head=(hw, hb))
        # a fresh buffer starts as NaN: in the gradient-keeping path the value is written by the BACKWARD launch, and a caller that
        # never runs it must not read a plausible-looking stale number (a caller-supplied `out` keeps its previous loss)
        loss = spec.out if spec.out is not None else torch.full((1,), float('nan'), dtype=torch.float32, device=x.device)
        _ResChain8Head.last_logits = logits
        if not keep:             # forward only: the loss now, by the stand-alone kernel (same arithmetic)
            K.critic_head_hinge(pooled, hw, hb, spec.n_real, spec.mode, want_dx=False, loss=loss, loss_scale=spec.loss_scale)
            return loss.detach() if spec.out is not None else loss
        ctx.cfg = (nb, True, Ws, bs)
        ctx.param_offset = 5
        ctx.kept = (x, h1s, ys)
        ctx.head = (head_W, head_b, hw, spec, pooled, logits, loss)
        return loss.detach() if spec.out is not None else loss

    @staticmethod
    def backward(ctx, g):
        head_W, head_b, hw, spec, pooled, logits, loss = ctx.head
        if _seed_scale.get(g.data_ptr()) != spec.loss_scale:
            raise NotImplementedError("the fused critic head differentiates the loss itself (loss.backward(gradient=grad_seed(loss, loss_scale)) "
                                      "with the loss scale its forward launch was given); use linear + hinge_*_loss for a weighted sum of losses")
        wt = bt = None
        ret = [None, None]
        if ctx.needs_input_grad[1]:
            wt, acc = _target(head_W)
            ret[0] = None if acc else wt
        if head_b is not None and ctx.needs_input_grad[2]:
            bt, bacc = _target(head_b)
            ret[1] = None if bacc else bt
        head = dict(logits=logits, w=hw, pooled=pooled, loss=loss, w_grad=wt.view(-1) if wt is not None else None, b_grad=bt,
                    n_real=spec.n_real, mode=spec.mode, loss_scale=spec.loss_scale)
        dx, grads = _res_chain8_grads(ctx, None, head=head)
        return (dx if ctx.needs_input_grad[0] else None, ret[0], ret[1], None, None, *grads)


def res_chain8(x, blocks_params, pool=False, head=None):
    """blocks_params: [(W1, b1, W2, b2), ...] (1 or 2 blocks).  head = (HingeHeadSpec, W [128,1], b [1] | None) with pool: the
    critic's last dense layer and its hinge loss inside the two launches -> the loss (logits ride along as `loss.logits`)"""
    flat = [t for bp in blocks_params for t in bp]
    todo = [w for w in flat[0::2] if getattr(w, "_prep_res", None) is None]
    if todo:                                         # not prepared by a batched pass (sn.precomputed)
        K.prep_weights_batched(todo, want_d=True, kinds=[4] * len(todo))
    if head is not None:
        assert pool
        spec, hW, hb = head
        loss = _ResChain8Head.apply(x, hW, hb, spec, torch.is_grad_enabled(), *flat)
        loss.logits, _ResChain8Head.last_logits = _ResChain8Head.last_logits, None
        return loss
    return _ResChain8.apply(x, pool, torch.is_grad_enabled(), *flat)


class _LinearSmall(Function):
    """Latency-sized dense layer straight on the fp32 weight (no MFMA operand preparation)."""

    @staticmethod
    def forward(ctx, x, W, bias):
        ctx.save_for_backward(x, W)
        ctx.bias = bias
        return K.linear_fwd(x, W.detach(), bias.detach() if bias is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        bias = ctx.bias
        g = _c(dy)
        dW = db = None
        wt = bt = None
        if ctx.needs_input_grad[1]:
            wt, acc = _target(W)
            dW = None if acc else wt
        if bias is not None and ctx.needs_input_grad[2]:
            bt, bacc = _target(bias)
            db = None if bacc else bt
        dx = K.linear_bwd(g, x, W.detach(), ctx.needs_input_grad[0], wt, bt)
        return dx, dW, db


SMALL_LINEAR_MACS = 1 << 24   # below this many multiply-adds a dense layer is a launch-latency problem


def linear(x, W, bias=None):
    """x [n, Cin] bf16, W fp32 [Cin, Cout]: small layers on dedicated kernels, large ones as the 1x1 case of the
    conv engine."""
    n = x.shape[0]
    if n * W.shape[0] * W.shape[1] <= SMALL_LINEAR_MACS:
        return _LinearSmall.apply(_c(x), W, bias)
    return _Conv2d.apply(x.view(n, 1, 1, x.shape[1]), W, bias, None, False, False, False, False).view(n, W.shape[1])


class _SpectralNorm(Function):
    """Batched spectral_normed_weight (common/ops/sn.py:15-69); backward = full gradient."""

    @staticmethod
    def forward(ctx, batch, *Ws):
        ctx.batch = batch
        outs = batch.forward()
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        join_wgrad()          # the dW_bar buffers are written by the filter-gradient stream
        batch = ctx.batch
        tgts, rets, gl = [], [], []
        for W, g, wb in zip(batch.weights, gs, batch.W_bar):
            tgt, acc = _target(W)
            tgts.append(tgt)
            rets.append(None if acc else tgt)
            gl.append(_c(g) if g is not None else torch.zeros_like(wb))
        opt = SN_DEFER_APPLY
        if (opt is not None and batch.state is not None and batch.inplace and batch.n <= 16 and all(r is None for r in rets)
                and all(w.shape[-1] <= 256 for w in batch.weights)):
            # the train step's optimiser launch applies the spectral norm's gradient itself (gank_sn_adam_fwd_a): only the
            # <dW_bar, W> partials now.  Until that launch the weights' gradient views hold NO spectral-norm contribution.
            batch.backward_gw(gl, tgts)
            opt.pending_sn = batch
        else:
            batch.backward(gl, tgts)
        return (None, *rets)


SN_DEFER_APPLY = None     # an optimiser object (attribute `pending_sn`) whose next launch applies the spectral norm's backward pass itself


class defer_sn_apply:
    """with defer_sn_apply(optimiser): backward passes inside leave the second half of the batched spectral-norm backward to the
    optimiser's launch (SNGANTrainer: AdamTF.apply -> gank_sn_adam_fwd_a).  None = off."""

    def __init__(self, opt):
        self.opt = opt

    def __enter__(self):
        global SN_DEFER_APPLY
        self.prev, SN_DEFER_APPLY = SN_DEFER_APPLY, self.opt
        if self.opt is not None:
            self.opt.pending_sn = None
        return self

    def __exit__(self, *exc):
        global SN_DEFER_APPLY
        SN_DEFER_APPLY = self.prev
        return False


def spectral_norm_batch(Ws, us, snapshot=False, inplace=False, prep=None, label=None, state=None):
    """Ws: fp32 weights (Cout last); us: fp32 [.., C] vectors READ by this call (pass snapshots if the
    stored u is overwritten before backward, or let the kernels keep one: snapshot=True; inplace=True writes
    u_final straight over `us`).  prep = (kinds, want_d): the bf16 MFMA operand copies of the normalised weights come
    out of the same launch pair (kernels.prep_weights_batched's kinds; they land on the W_bar tensors).  label = (embedding
    table, index of the dense weight in Ws, bias | None): the per-label rows of that dense layer too (`W_bar._label_T`).
    Returns (W_bars tuple, SnBatch)."""
    batch = K.SnBatch(list(Ws), [u.detach() for u in us], snapshot, inplace, state=state)
    batch.prep, batch.label = prep, label
    outs = _SpectralNorm.apply(batch, *Ws)
    for o, wb in zip(outs, batch.W_bar):      # autograd hands out fresh tensor objects: carry the operand copies over
        for attr in ("_prep", "_prep_up", "_prep_pool", "_prep_res", "_prep_cpres", "_label_T"):
            val = getattr(wb, attr, None)
            if val is not None and o is not wb:
                setattr(o, attr, val)
    return outs, batch


CBN_REMASK = True   # relu mask of the backward pass recomputed from x instead of read from y


class _CondBatchNorm(Function):
    @staticmethod
    def forward(ctx, x, labels, gamma, beta, groups, relu):
        cs = getattr(x, "_cbn_stats", None)
        if cs is not None and cs.groups == groups and cs.sums.shape[-1] == x.shape[-1]:
            y, stats = K.cbn_fwd_from_sums(x, labels, gamma.detach(), beta.detach(), cs, relu)      # statistics came with x
        else:
            y, stats = K.cbn_fwd(x, labels, gamma.detach(), beta.detach(), groups, relu)
        ctx.save_for_backward(x, y, labels, gamma, beta, stats)
        ctx.cfg = (groups, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, labels, gamma, beta, stats = ctx.saved_tensors
        groups, relu = ctx.cfg
        tg, accg = _target(gamma)
        tb, accb = _target(beta)
        dx = K.cbn_bwd(_c(dy), x, y, labels, gamma.detach(), stats, tg, tb, groups, relu, beta=beta.detach() if CBN_REMASK else None)
        return dx, None, (None if accg else tg), (None if accb else tb), None, None


def cond_batchnorm(x, labels, gamma, beta, groups=1, relu=False):
    return _CondBatchNorm.apply(x, labels, gamma, beta, groups, relu)


class _BatchNormStats(Function):
    """cond_batchnorm with the batch statistics as a second, non-differentiable output ([groups, 2, C]: mean, invstd)."""

    @staticmethod
    def forward(ctx, x, labels, gamma, beta, groups, relu, eps=1e-5):
        y, stats = K.cbn_fwd(x, labels, gamma.detach(), beta.detach(), groups, relu, eps)
        ctx.save_for_backward(x, y, labels, gamma, beta, stats)
        ctx.cfg = (groups, relu)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # the statistics output has no gradient: autograd must not fill a zero tensor for it
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return None, None, None, None, None, None, None
        x, y, labels, gamma, beta, stats = ctx.saved_tensors
        groups, relu = ctx.cfg
        tg, accg = _target(gamma)
        tb, accb = _target(beta)
        dx = K.cbn_bwd(_c(dy), x, y, labels, gamma.detach(), stats, tg, tb, groups, relu)
        return dx, None, (None if accg else tg), (None if accb else tb), None, None, None


def batchnorm_with_stats(x, labels, gamma, beta, groups=1, relu=False, eps=1e-5):
    return _BatchNormStats.apply(x, labels, gamma, beta, groups, relu, eps)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, stats = K.layer_norm_fwd(x, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        tg, accg = _target(gamma)
        tb, accb = _target(beta)
        dx = K.layer_norm_bwd(_c(dy), x, gamma.detach(), stats, tg, tb)
        return dx, (None if accg else tg), (None if accb else tb), None


def layer_norm(x, gamma, beta, eps=1e-12):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, x, eps):
        ctx.save_for_backward(x)
        ctx.eps = eps
        return K.pixel_norm_fwd(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.pixel_norm_bwd(_c(dy), x, ctx.eps), None


def pixel_norm(x, eps=1e-8):
    return _PixelNorm.apply(x, eps)


class _Fork(Function):
    """Explicit activation fan-out: two aliases forward, one add kernel backward."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)     # a branch whose gradient is None must not arrive as a zero tensor + an add
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        return K.add(_c(ga), _c(gb))


def fork(x):
    if not x.requires_grad:
        return x, x
    a, b = _Fork.apply(x)
    cs = getattr(x, "_cbn_stats", None)
    if cs is not None:                       # statistics that came with x (conv epilogue) belong to both aliases
        a._cbn_stats = b._cbn_stats = cs
    return a, b


class _ForkPool(Function):
    """Fan-out into (x, mean_pool2x2(x)) -- the main path and the pooled shortcut of a down-sampling block.  Backward: the
    pooled branch's gradient is unpooled INTO the main branch's (gank_unpool2x2_add with a base): one launch and three tensor
    passes where the separate fork + pool spent an unpool and an add (two launches, six passes)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), K.pool2x2(x, 0.25)

    @staticmethod
    def backward(ctx, ga, gp):
        if gp is None:
            return ga
        return K.unpool2x2_add(_c(gp), None if ga is None else _c(ga), 0.25)


def fork_pool(x):
    """-> (x for the main path, mean_pool2x2(x) for the shortcut)"""
    if not x.requires_grad:
        return x, K.pool2x2(x, 0.25)
    return _ForkPool.apply(x)


class _ForkPoolConv1x1(Function):
    """Fan-out of a 3-channel image into (x for the main path, conv1x1(mean_pool2x2(x)) + bias): the first critic block's
    shortcut (MeanPoolConv 1x1, gan_cifar_resnet.py:218-221) with the pool inside the conv's gather -- one launch for
    fork_pool + conv.  Backward = the same launches as the separate ops: filter gradient on the pooled image (a side
    output of the forward kernel), and, when the image itself needs a gradient (the generator update), the 1x1 input
    gradient unpooled into the main branch's."""
    last_sink = None

    @staticmethod
    def forward(ctx, x, W, bias, conv1=None):
        ctx.set_materialize_grads(False)
        cin, cout = W.shape[-2], W.shape[-1]
        wf, _ = _prepared(W, 1, cin, cout, True, False)
        keep = ctx.needs_input_grad[1]
        y1 = None
        if conv1 is not None:      # (wf1, bias1 | None, cout1): the 3x3 conv the caller applies to x next rides on the same launch
            y1, y, pooled = K.image_conv_pair_fprop(x, conv1[0], conv1[1], conv1[2], wf, bias.detach() if bias is not None else None, cout, keep_pooled=keep)
        else:
            y, pooled = K.meanpool_conv1x1_fprop(x, wf, bias.detach() if bias is not None else None, cout, keep_pooled=keep)
        ctx.save_for_backward(W, pooled)
        ctx.cfg = (cin, cout, bias)
        ctx.sink = _ForkPoolConv1x1.last_sink = None
        if FUSE_IMAGE_WGRAD and keep and not ctx.needs_input_grad[0] and cin == 3 and pooled is not None:
            ctx.sink = _ForkPoolConv1x1.last_sink = ImageWgradSink(pooled, W, bias)      # see ImageWgradSink: the ConvMeanPool this shortcut is added to may serve it
        xv = x.view_as(x)
        nd = [] if ctx.needs_input_grad[0] else [xv]
        # autograd marks EVERY output of a node differentiable when any input is (here: the weight); the alias of an image
        # that needs no gradient (the critic update) must not make the next conv compute an input gradient
        if y1 is not None:
            nd.append(y1)          # the VALUE of the caller's conv node (fork_pool_conv1x1 parks it on xv)
        if nd:
            ctx.mark_non_differentiable(*nd)
        return (xv, y) if y1 is None else (xv, y, y1)

    @staticmethod
    def backward(ctx, ga, gs, *_unused):
        out = _ForkPoolConv1x1._backward(ctx, ga, gs)
        return out + (None,)

    @staticmethod
    def _backward(ctx, ga, gs):
        W, pooled = ctx.saved_tensors
        cin, cout, bias = ctx.cfg
        dW = db = None
        if ctx.sink is not None and ctx.sink.done:
            tgt, acc, btgt, bacc = ctx.sink.tgt
            return ga, (None if acc else tgt), (None if (btgt is None or bacc) else btgt)
        if gs is None:
            return ga, None, None
        g = _c(gs)
        h, w = g.shape[1], g.shape[2]
        btgt = None
        if bias is not None and ctx.needs_input_grad[2]:
            btgt, bacc = _target(bias)
            db = None if bacc else btgt
        if ctx.needs_input_grad[1]:
            tgt, acc = _target(W)
            if narrow_wgrad_ok(cin, cout, 1, True):
                _defer_narrow(pooled, g, tgt, btgt, (h, w), 1)
            else:
                K.conv2d_wgrad(pooled, g, tgt, (h, w), 1, 0, 1.0, dbias=btgt)
            dW = None if acc else tgt
        elif btgt is not None:
            K.colsum(g, btgt, 1.0)
        dx = ga
        if ctx.needs_input_grad[0]:
            _, wd = _prepared(W, 1, cin, cout, False, True)
            dp = K.conv2d_dgrad(g, wd, (h, w), cin, 1)
            dx = K.unpool2x2_add(dp, None if ga is None else _c(ga), 0.25)
        return dx, dW, db


IMAGE_CONV_PAIR = True      # the critic's two image-side convs (conv_1 3x3 and the pooled 1x1 shortcut) in one launch


def fork_pool_conv1x1(x, W, bias=None, W1=None, b1=None):
    """-> (x for the main path, conv1x1(mean_pool2x2(x)) + bias); x [N,H,W,3], W fp32 [1,1,3,Cout] (or [3,Cout]).
    W1 [3,3,3,Cout1], b1: the filter / bias of the 3x3 conv the caller applies to the returned x next: its VALUE comes from the same
    launch and is parked on that tensor (`_conv_result`), where conv2d() with exactly these tensors picks it up (the conv stays a
    node of its own with its own backward)."""
    conv1 = None
    if (IMAGE_CONV_PAIR and W1 is not None and W1.dim() == 4 and W1.shape[0] == 3 and W1.shape[2] == 3 and W1.shape[3] % 128 == 0
            and W.shape[-1] % 128 == 0 and x.is_cuda):
        wf1, _ = _prepared(W1, 3, 3, W1.shape[3], True, False)
        conv1 = (wf1, b1.detach() if b1 is not None else None, W1.shape[3])
    if conv1 is not None:
        xv, y, y1 = _ForkPoolConv1x1.apply(x, W, bias, conv1)
        xv._conv_result = (W1, b1, y1)
    else:
        xv, y = _ForkPoolConv1x1.apply(x, W, bias)
    if _ForkPoolConv1x1.last_sink is not None:
        y._shortcut_sink, _ForkPoolConv1x1.last_sink = _ForkPoolConv1x1.last_sink, None
    return xv, y


def fork_pool_conv1x1_ok(x, cout):
    return x.dim() == 4 and x.shape[3] == 3 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and cout % 128 == 0 and x.is_cuda


class _Relu(Function):
    @staticmethod
    def forward(ctx, x, leak):
        ctx.save_for_backward(x)
        ctx.leak = leak
        return K.relu_fwd(x, leak)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.relu_bwd(_c(dy), x, ctx.leak), None


def relu(x, leak=0.0):
    return _Relu.apply(x, leak)


class _MeanPool(Function):
    @staticmethod
    def forward(ctx, x):
        return K.pool2x2(x, 0.25)

    @staticmethod
    def backward(ctx, dy):
        return K.unpool2x2_add(_c(dy), None, 0.25)


def meanpool2x2(x):
    return _MeanPool.apply(x)


class _Upsample(Function):
    @staticmethod
    def forward(ctx, x):
        return K.unpool2x2_add(x, None, 1.0)

    @staticmethod
    def backward(ctx, dy):
        return K.pool2x2(_c(dy), 1.0)


def upsample_nn2x(x):
    return _Upsample.apply(x)


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _ReluMeanPoolHW(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.relu_meanpool_hw_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.relu_meanpool_hw_bwd(_c(dy), x)


def relu_meanpool_hw(x):
    return _ReluMeanPoolHW.apply(x)


class _ConcatTile(Function):
    @staticmethod
    def forward(ctx, a, e):
        ctx.c1 = a.shape[3]
        return K.concat_tile_fwd(a, e)

    @staticmethod
    def backward(ctx, dy):
        da, de = K.concat_tile_bwd(_c(dy), ctx.c1)
        return da, de


def concat_tile(a, e):
    return _ConcatTile.apply(a, e)


class _ConcatLabel(Function):
    """tf.concat([a, tile(Linear(embed_y(labels)))], 3) (gan_cifar_resnet.py:276-284) through the per-label table
    T [V, C2] = bf16(bf16(table) W + bias): the dense layer runs on the V table rows instead of the N samples (T rides on W
    as `W._label_T` when the batched spectral norm made it, else one launch builds it here), the concat gathers T by label.
    Backward: da = dy[..., :C1]; the tiled half summed over pixels per sample in fp32, then per label, then the dense
    layer's and the table's gradients from the V-row sums -- 2 launches for what was 4 (concat, dense dx, dense dw, embedding)."""

    @staticmethod
    def forward(ctx, a, labels, table, W, bias):
        T = getattr(W, "_label_T", None)
        if T is None:
            T = K.label_dense_table(table.detach(), W.detach(), bias.detach() if bias is not None else None)
        ctx.save_for_backward(labels, table, W)
        ctx.bias = bias
        ctx.c1 = a.shape[3]
        return K.concat_label_fwd(a, T, labels)

    @staticmethod
    def backward(ctx, dy):
        labels, table, W = ctx.saved_tensors
        bias = ctx.bias
        da, de32 = K.concat_label_bwd(_c(dy), ctx.c1)
        dt, dw, db = _label_dense_grads(ctx, de32, labels, table, W, bias)
        return (da if ctx.needs_input_grad[0] else None), None, dt, dw, db


def concat_label(a, labels, table, W, bias=None):
    return _ConcatLabel.apply(a, labels, table, W, bias)


def _label_dense_grads(ctx, de32, labels, table, W, bias, defer=False):
    """the dense layer's / table's gradients from the per-sample sums of the tiled half (shared by both concat forms).
    defer: -> (dt, dw, db, finish): the targets are handed to autograd now, finish(parts) issues the launch later (in stream order
    before anything reads them: join_wgrad)"""
    need_t, need_w = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
    need_b = bias is not None and ctx.needs_input_grad[4]
    dt = dw = db = None
    tt = wt = bt = None
    if need_t:
        tt, acc = _target(table)
        dt = None if acc else tt
    if need_w:
        wt, acc = _target(W)
        dw = None if acc else wt
    if need_b:
        bt, acc = _target(bias)
        db = None if acc else bt

    def finish(rows):
        if need_t or need_w or need_b:
            if labels is None:                     # rows summed per label already, fp32 [parts, V, C2]
                K.label_dense_bwd_parts(rows, table.detach(), W.detach(), wt, bt, tt)
            else:
                K.label_dense_bwd(rows, labels, table.detach(), W.detach(), wt, bt, tt)
    if defer:
        return dt, dw, db, finish
    finish(de32)
    return dt, dw, db


class _ConcatLabelForkPool(Function):
    """_ConcatLabel followed by the fan-out of a down-sampling residual block (fork_pool): -> (y, mean_pool2x2(y)) from one
    forward launch (the tiled half's pool is the table row itself); backward: the two branch gradients meet in ONE launch that
    writes da and the per-sample sums of the tiled half -- the unpool-add and the concat split never materialise dy."""

    @staticmethod
    def forward(ctx, a, labels, table, W, bias):
        ctx.set_materialize_grads(False)
        T = getattr(W, "_label_T", None)
        if T is None:
            T = K.label_dense_table(table.detach(), W.detach(), bias.detach() if bias is not None else None)
        ctx.save_for_backward(labels, table, W)
        ctx.bias = bias
        ctx.c1 = a.shape[3]
        return K.concat_label_pool_fwd(a, T, labels)

    @staticmethod
    def backward(ctx, gy, gp):
        labels, table, W = ctx.saved_tensors
        if gp is None:
            da, de32 = K.concat_label_bwd(_c(gy), ctx.c1)
        else:
            da, de32 = K.concat_label_unpool_bwd(None if gy is None else _c(gy), _c(gp), ctx.c1)
        dt, dw, db = _label_dense_grads(ctx, de32, labels, table, W, ctx.bias)
        return (da if ctx.needs_input_grad[0] else None), None, dt, dw, db


POOLED_LABEL_PART = True    # ... what is left of the join launch (the tiled vector's gradient through the pooled branch) as a tenth part of the label-gradient launch
JOIN_IN_DGRAD = True        # ... the pooled shortcut's gradient joins the feature gradient in the input-gradient launch's epilogue
LABEL_BWD_RIDER = False     # ... and its label gradients as extra workgroups of the input-gradient launch (measured: 20.7 + 6.8 us -> 26.6 us, no gain)
TAP_SUMS_RIDER = True       # ... its per-label tap sums as extra workgroups of the feature half's filter-gradient launch
FACTOR_LABEL_CONV = True    # D.Block.2.Conv1 with the tiled (spatially constant) half of its input factored out (csrc/label_conv.hip, round 5)


class _ConcatLabelConv1(Function):
    """concat(a, tile(T[labels])) (gan_cifar_resnet.py:282-284) -> the fan-out of D.Block.2 with conv_1 of the main path
    (relu -> 3x3 conv, :186-190) computed WITHOUT the concatenated tensor: -> (h1 = conv_1(relu(concat)) [N,16,16,Cout],
    mean_pool2x2(concat) [N,8,8,C1+C2] for the pooled shortcut).  The tiled half of the input is one vector per sample, so its
    contribution to conv_1 is a per-(label, border class) bias row (kernels.label_conv3x3_table) and the conv itself runs on the
    C1 feature channels: half the multiply-adds in the forward pass, the input gradient and the filter gradient.  Backward: the
    feature half's filter gradient (a slab job into its rows of the gradient), the tiled half's filter gradient and the gradient of
    the tiled vector from nine per-tap sums of dh1 (kernels.label_conv3x3_bwd), the feature input gradient; both branch gradients
    then meet as in _ConcatLabelForkPool."""

    @staticmethod
    def forward(ctx, a, labels, table, W_emb, b_emb, W1, b1, shortcut):
        ctx.set_materialize_grads(False)
        T = getattr(W_emb, "_label_T", None)
        if T is None:
            T = K.label_dense_table(table.detach(), W_emb.detach(), b_emb.detach() if b_emb is not None else None)
        c1, cout = a.shape[3], W1.shape[3]
        rf, rd = W1._prep_feat
        sc = None
        if shortcut is not None:       # (ws_f, bias_s | None, Cs): the block's 1x1 shortcut conv on the pooled concatenation rides on the same launch
            bt, lists, yp, sc = K.label_conv3x3_table_pooled(W1.detach(), c1, T, b1.detach() if b1 is not None else None, labels, a, shortcut)
        else:
            bt, lists, yp = K.label_conv3x3_table_pooled(W1.detach(), c1, T, b1.detach() if b1 is not None else None, labels, a)     # one launch
        h1 = K.img16_conv3x3_label_bias(a, rf, bt, labels, cout, K.IN_RELU)
        ctx.save_for_backward(a, labels, table, W_emb, W1, T, lists)
        ctx.b_emb, ctx.b1, ctx.rd = b_emb, b1, rd
        if sc is None:
            return h1, yp
        ctx.mark_non_differentiable(sc)        # the value of a conv node of its own (whose backward is unchanged): see concat_label_conv1
        return h1, yp, sc

    @staticmethod
    def backward(ctx, dh1, gp, *_unused):
        out = _ConcatLabelConv1._backward(ctx, dh1, gp)
        return out + (None,)

    @staticmethod
    def _backward(ctx, dh1, gp):
        a, labels, table, W_emb, W1, T, lists = ctx.saved_tensors
        n, c1, cout = a.shape[0], a.shape[3], W1.shape[3]
        dW = db = None
        parts = da = None
        joined = False
        need_label = ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or (ctx.b_emb is not None and ctx.needs_input_grad[4])
        if dh1 is not None:
            g = _c(dh1)
            btgt = None
            if ctx.b1 is not None and ctx.needs_input_grad[6]:
                btgt, bacc = _target(ctx.b1)
                db = None if bacc else btgt
            if ctx.needs_input_grad[5]:
                tgt, acc = _target(W1)
                tgt4 = tgt.view(3, 3, W1.shape[2], cout)
                if SLAB_WGRADS and BATCH_SMALL_WGRADS and K.conv2d_wgrad_rows_ok(n, (16, 16), c1, cout, 3, K.IN_RELU):
                    # (summed with the pass's other slabs; the per-label tap sums of dh1 ride on the same launch)
                    sums = K.conv2d_wgrad_rows(a, g, tgt4, (16, 16), 3, K.IN_RELU, _slab_jobs, dbias=btgt,
                                               tap_sums=(lists, T.shape[0]) if TAP_SUMS_RIDER else None)
                    if sums is not None and LABEL_BWD_RIDER and ctx.needs_input_grad[0]:
                        # ... and the label gradients on the input-gradient launch
                        da, parts = K.img16_conv3x3_label_bwd(g, ctx.rd, a, c1, sums, T, W1.detach(), c1, tgt4)
                    elif sums is not None and POOLED_LABEL_PART and JOIN_IN_DGRAD and gp is not None and need_label and ctx.needs_input_grad[0]:
                        # ... with the pooled branch's share of the tiled vector's gradient as a tenth part (extra workgroups): the join
                        # launch below has nothing left to do
                        label = (sums, lists, T, W1.detach(), c1, tgt4, _c(gp), c1, n)
                        if LABEL_BWD_IN_SUM_SLABS and BATCH_SMALL_WGRADS:
                            # ... and the launch itself rides on the pass's slab-summing launch (join_wgrad), the dense layer's gradients behind it
                            # (their targets go to autograd now, as a deferred filter gradient's do)
                            dt_, dw_, dbe_, finish = _label_dense_grads(ctx, None, None, table, W_emb, ctx.b_emb, defer=True)
                            _label_jobs.append((label, finish))
                            parts = ("deferred", dt_, dw_, dbe_)
                        else:
                            parts = K.label_conv3x3_bwd_pooled(*label)
                    else:
                        parts = K.label_conv3x3_bwd(g, lists, T, W1.detach(), c1, tgt4, sums=sums)
                else:                          # small batches: through a zero-filled staging buffer, merged by the same launch
                    tmp = K.zeros_f32((3, 3, c1, cout), a.device)
                    K.conv2d_wgrad(a, g, tmp, (16, 16), 3, K.IN_RELU, 1.0, dbias=btgt)
                    parts = K.label_conv3x3_bwd(g, lists, T, W1.detach(), c1, tgt4, tmp)
                dW = None if acc else tgt
            else:
                if btgt is not None:
                    K.colsum(g, btgt, 1.0)
                if need_label:                 # (a frozen filter with a trainable label branch: no such graph in the train steps)
                    scratch = K.zeros_f32(tuple(W1.shape), a.device)
                    parts = K.label_conv3x3_bwd(g, lists, T, W1.detach(), c1, scratch)
            if ctx.needs_input_grad[0] and da is None:
                if gp is not None and JOIN_IN_DGRAD:
                    # ... and the pooled branch's gradient joins in the same epilogue: 0.25 * unpool(gp[..., :C1]) (no join pass over da)
                    da = K.img16_conv3x3_dgrad_unpool(g, ctx.rd, a, _c(gp), c1, 0.25)
                    joined = True
                else:
                    da = K.img16_conv3x3(g, ctx.rd, None, c1, 0, a)      # relu mask of the feature half in the epilogue
        if gp is not None and joined and isinstance(parts, tuple):        # deferred (join_wgrad issues the launches)
            return da, None, parts[1], parts[2], parts[3], dW, db
        if gp is not None and joined and parts is not None and parts.shape[0] == 10:
            dt, dw, dbe = _label_dense_grads(ctx, parts, None, table, W_emb, ctx.b_emb)
            return da, None, dt, dw, dbe, dW, db
        if gp is not None:
            if da is None:
                da = K.zeros_bf16(tuple(a.shape), a.device) if hasattr(K, "zeros_bf16") else torch.zeros_like(a)
            if joined and not need_label:
                de32 = None                    # (the generator update: nobody asks for the tiled vector's gradient -- no launch)
            elif joined:
                _, de32 = K.concat_label_unpool_bwd_factored(c1, _c(gp), parts if need_label else None, labels, lists)
            else:
                da, de32 = K.concat_label_unpool_bwd_factored(da, _c(gp), parts if need_label else None, labels, lists)
        else:                                  # (no pooled branch: not a graph of this library; the per-label sums at each label's first sample)
            de32 = torch.zeros((n, T.shape[1]), dtype=torch.float32, device=a.device)
            if parts is not None:
                pl = parts.sum(0)
                for v in range(T.shape[0]):
                    if int(lists[v, 0]) > 0:
                        de32[int(lists[v, 1])] = pl[v]
        dt = dw = dbe = None
        if need_label:
            dt, dw, dbe = _label_dense_grads(ctx, de32, labels, table, W_emb, ctx.b_emb)
        return (da if ctx.needs_input_grad[0] else None), None, dt, dw, dbe, dW, db


SHORTCUT_IN_TABLE_LAUNCH = True     # ... the block's 1x1 shortcut conv on the pooled concatenation computed by the table / pooling launch


def concat_label_conv1(a, labels, table, W_emb, b_emb, W1, b1, Ws=None, bs=None):
    """-> (conv3x3(relu(concat(a, tile(T[labels]))), W1) + b1, mean_pool2x2(concat)): see _ConcatLabelConv1.
    Ws [1,1,C1+C2,Cs], bs: the filter / bias of the 1x1 conv the caller applies to the pooled result next (the down-sampling block's
    shortcut): its VALUE is computed by the same launch and parked on the pooled tensor (`_conv_result`), where conv2d() with exactly
    these tensors picks it up instead of launching -- the conv stays a node of its own, its backward is unchanged."""
    sc_arg = None
    if SHORTCUT_IN_TABLE_LAUNCH and Ws is not None and Ws.dim() == 4 and Ws.shape[0] == 1:
        prep = getattr(Ws, "_prep", None)
        c2 = W1.shape[2] - a.shape[3]
        if prep is not None and prep[0] is not None and K.label_conv3x3_table_pooled_shortcut_ok(a, c2, prep[0], Ws.shape[3]) and Ws.shape[2] == W1.shape[2]:
            sc_arg = (prep[0], bs.detach() if bs is not None else None, Ws.shape[3])
    if sc_arg is None:
        return _ConcatLabelConv1.apply(a, labels, table, W_emb, b_emb, W1, b1, None)
    h1, yp, sc = _ConcatLabelConv1.apply(a, labels, table, W_emb, b_emb, W1, b1, sc_arg)
    yp._conv_result = (Ws, bs, sc)
    return h1, yp


def concat_label_conv1_ok(a, W1):
    """16x16 feature maps, a filter whose first half of input channels is `a`, and the sliced operands attached by the batched preparation"""
    return (FACTOR_LABEL_CONV and a.is_cuda and a.dim() == 4 and tuple(a.shape[1:3]) == (16, 16) and W1.dim() == 4 and W1.shape[0] == 3
            and W1.shape[2] == 2 * a.shape[3] and getattr(W1, "_prep_feat", None) is not None and W1._prep_feat[1] is not None
            and K.img16_conv3x3_ok(a.shape[0], (16, 16), a.shape[3], W1.shape[3]) and W1.shape[3] % 8 == 0 and 256 % (W1.shape[3] // 8) == 0
            and a.shape[3] % 16 == 0)


def concat_label_fork_pool(a, labels, table, W, bias=None):
    """-> (concat(a, tile(T[labels])), its 2x2 mean): input pair of a down-sampling ResidualBlock(prefork=...)"""
    return _ConcatLabelForkPool.apply(a, labels, table, W, bias)


class _Embedding(Function):
    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(table, idx)
        return K.embedding_fwd(table.detach(), idx)

    @staticmethod
    def backward(ctx, dy):
        table, idx = ctx.saved_tensors
        tgt, acc = _target(table)
        K.embedding_bwd(_c(dy), idx, tgt)
        return (None if acc else tgt), None


def embedding(table, idx):
    return _Embedding.apply(table, idx)


class _Loss(Function):
    """loss value (fp32[1]) with d loss / d logits computed in the same forward launch."""

    @staticmethod
    def forward(ctx, logits, kind, arg, out=None):
        # out: a _Box around a persistent fp32[1] buffer the value is written to (autograd must not see that tensor as an
        # input); the result is a fresh alias of it
        buf = out.t if out is not None else None
        if kind == "hinge_d":
            loss, dl, dl32 = K.hinge_d_loss(logits, arg, buf)
        elif kind == "wgan_d":
            loss, dl, dl32 = K.wgan_d_loss(logits, arg, buf)
        elif kind == "hinge_g":
            loss, dl, dl32 = K.hinge_g_loss(logits, buf)
        elif kind == "xent":
            loss, dl, dl32 = K.softmax_xent(logits, arg)
        elif kind.startswith("pointwise"):
            loss, dl, dl32 = K.gan_pointwise_loss(logits, arg[0], arg[1], buf)
        else:
            raise NotImplementedError(kind)
        ctx.save_for_backward(dl, dl32)
        return loss.detach() if buf is not None else loss

    @staticmethod
    def backward(ctx, g):
        dl, dl32 = ctx.saved_tensors
        # g = d(total)/d(loss), fp32[1].  The train step differentiates the loss itself with a persistent unit seed
        # (unit_seed): the bf16 gradient of the forward launch is returned as is.  Any other upstream gradient (a
        # weighted sum of losses, loss / accum_steps) scales the fp32 gradient on the device and rounds once.
        if g.data_ptr() in _unit_seed_ptrs:
            return dl, None, None, None
        return K.loss_grad_scale(dl32, _c(g.to(torch.float32)).reshape(1)), None, None, None


_unit_seeds = {}
_unit_seed_ptrs = set()
_seed_scale = {}          # data_ptr of a registered seed tensor -> its value (1.0, or the static loss scale)


def grad_seed(loss, scale=1.0):
    """Persistent gradient seed for `loss.backward(gradient=...)`: no ones_like fill per update.  scale = 1: _Loss / the fused
    head recognise it (by identity) and hand out the 16-bit gradient of the forward launch as is.  scale = a power of two
    (static loss scaling, the fp16 build): the loss nodes multiply d loss / d logits by it before rounding to 16 bits -- every
    gradient behind them is `scale` times larger and the optimiser's grad_scale divides it out."""
    key = (loss.device, tuple(loss.shape), float(scale))
    s = _unit_seeds.get(key)
    if s is None:
        s = _unit_seeds[key] = torch.full_like(loss, float(scale))      # never freed, never written: its address identifies it
        _seed_scale[s.data_ptr()] = float(scale)
        if float(scale) == 1.0:
            _unit_seed_ptrs.add(s.data_ptr())
    return s


_constants = {}


def constant_like(t, value):
    """a persistent tensor of t's shape / dtype / device filled with `value` ONCE (grad_outputs of an autograd.grad call, ...):
    never written again, so a captured graph may read it"""
    key = (t.device, tuple(t.shape), t.dtype, float(value))
    c = _constants.get(key)
    if c is None:
        c = _constants[key] = torch.full_like(t, float(value))
    return c


def unit_seed(loss):
    """Persistent all-ones gradient seed (grad_seed with scale 1)."""
    return grad_seed(loss, 1.0)


class _HingeHead(Function):
    """D.Output (a dense layer to one logit) + hinge loss in one launch (kernels.critic_head_hinge).  The weight and bias
    gradients are accumulated into their targets by the forward launch (a backward pass always follows in the train step);
    the loss must be differentiated directly (unit upstream gradient) -- a weighted sum goes through linear + hinge_*_loss."""

    @staticmethod
    def forward(ctx, x, W, bias, n_real, mode, out, loss_scale=1.0):
        ctx.loss_scale = float(loss_scale)
        need_w = ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        wt = bt = None
        ctx.ret = [None, None]
        if need_w:
            wt, acc = _target(W)
            ctx.ret[0] = None if acc else wt
        if need_b:
            bt, bacc = _target(bias)
            ctx.ret[1] = None if bacc else bt
        for prm, tgt in ((W, wt), (bias, bt)):
            # this launch accumulates into the flat gradient buffer DURING THE FORWARD pass: a forward-only call (a loss
            # evaluation, a capture that aborts before its backward pass) must not leave the buffer marked clean
            fl = getattr(prm, "_flat", None) if tgt is not None else None
            if fl is not None:
                fl["clean"] = False
        buf = out.t if out is not None else None
        loss, logits, dx = K.critic_head_hinge(_c(x), W.detach().reshape(-1), bias.detach() if bias is not None else None, n_real, mode,
                                               ctx.needs_input_grad[0], wt.view(-1) if wt is not None else None, bt, buf, loss_scale=ctx.loss_scale)
        ctx.dx = dx
        _HingeHead.last_logits = logits
        return loss.detach() if buf is not None else loss

    @staticmethod
    def backward(ctx, g):
        if _seed_scale.get(g.data_ptr()) != ctx.loss_scale:
            raise NotImplementedError("the fused critic head differentiates the loss itself (loss.backward(gradient=grad_seed(loss, loss_scale)) "
                                      "with the loss scale its forward launch was given); use linear + hinge_*_loss for a weighted sum of losses")
        return ctx.dx, ctx.ret[0], ctx.ret[1], None, None, None, None


def hinge_d_head(x, W, bias, n_real, out=None, loss_scale=1.0):
    """hinge_d_loss(linear(x, W, bias), n_real) in one launch; the logits ride along as `loss.logits` (bf16 [M], detached).
    loss_scale: see grad_seed (the backward seed must carry the same scale)"""
    loss = _HingeHead.apply(x, W, bias, int(n_real), 0, _Box(out) if out is not None else None, float(loss_scale))
    loss.logits, _HingeHead.last_logits = _HingeHead.last_logits, None
    return loss


def hinge_g_head(x, W, bias, out=None, loss_scale=1.0):
    loss = _HingeHead.apply(x, W, bias, 0, 1, _Box(out) if out is not None else None, float(loss_scale))
    loss.logits, _HingeHead.last_logits = _HingeHead.last_logits, None
    return loss


class _Box:
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class _WeightedSum(Function):
    """sum_i w_i * term_i of fp32 scalar losses (one launch).  Backward: a unit seed passes through a weight of 1 untouched (the
    loss nodes recognise it and hand out the gradient their forward launch made); any other weight turns it into the persistent
    constant seed of that value (grad_seed: no launch either); an upstream gradient that is no seed is scaled on the device."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights = tuple(float(w) for w in weights)
        return K.weighted_sum_f32([_c(t) for t in terms], ctx.weights)

    @staticmethod
    def backward(ctx, g):
        sc = _seed_scale.get(g.data_ptr())
        outs = []
        for w in ctx.weights:
            if sc is not None:
                outs.append(g if w == 1.0 else grad_seed(g, sc * w))
            else:
                outs.append(K.weighted_sum_f32([_c(g)], [w]))
        return (None,) + tuple(outs)


def weighted_sum(terms, weights=None):
    """the train steps' sums of loss terms (d_loss = gan + gp + ac; gen_loss = gan_weight * GAN + l1_weight * L1) without framework arithmetic"""
    weights = [1.0] * len(terms) if weights is None else weights
    return _WeightedSum.apply(tuple(weights), *terms)


class _ConcatRows(Function):
    """tf.concat([a, b], axis=0) of two contiguous tensors: two device copies of the library (captured graphs hold kernel nodes
    only); backward: the two row ranges of the gradient (views)"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.na = a.shape[0]
        out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        K.copy_(out[:ctx.na], a)
        K.copy_(out[ctx.na:], b)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.na], g[ctx.na:]


def concat_rows(a, b):
    return _ConcatRows.apply(a, b)


def hinge_d_loss(logits, n_real, out=None):
    """out: persistent fp32[1] buffer that also receives the loss value (no copy launch for the reported loss)"""
    return _Loss.apply(logits, "hinge_d", n_real, _Box(out) if out is not None else None)


def hinge_g_loss(logits, out=None):
    return _Loss.apply(logits, "hinge_g", None, _Box(out) if out is not None else None)


def wgan_d_loss(logits, n_real, out=None):
    return _Loss.apply(logits, "wgan_d", n_real, _Box(out) if out is not None else None)


def softmax_xent(logits, labels):
    return _Loss.apply(logits, "xent", labels)


def gan_pointwise_loss(logits, n_real, kind, out=None):
    """the least-squares / sigmoid-cross-entropy / minimax branches of get_loss (kernels.gan_pointwise_loss)"""
    return _Loss.apply(logits, "pointwise", (int(n_real), int(kind)), _Box(out) if out is not None else None)


class _Cast(Function):
    @staticmethod
    def forward(ctx, x, to_bf16):
        ctx.to_bf16 = to_bf16
        return K.to_bf16(x) if to_bf16 else K.to_f32(x)

    @staticmethod
    def backward(ctx, g):
        return (K.to_f32(_c(g)) if ctx.to_bf16 else K.to_bf16(_c(g))), None


def to_bf16(x):
    return x if x.dtype == BF16 else _Cast.apply(x, True)


def to_f32(x):
    return x if x.dtype == torch.float32 else _Cast.apply(x, False)


# ---------------------------------------------------------------- PGGAN / Pix2Pix operators
class _Blend(Function):
    """(1 - alpha) * a + alpha * b: the fade-in of a new resolution (PGGAN/model_nvidia.py:116,206); alpha is a Python float
    or an fp32[1] device tensor (a placeholder fed per step in the reference, train.py:81)"""

    @staticmethod
    def forward(ctx, a, b, alpha):
        if torch.is_tensor(alpha):          # fp32[1] on the device: read by the kernels (a captured step replays with today's value)
            ctx.alpha = alpha
            return K.blend_dev(_c(a), _c(b), alpha, 0)
        ctx.alpha = float(alpha)
        return K.axpby(a, b, 1.0 - ctx.alpha, ctx.alpha)

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        if torch.is_tensor(ctx.alpha):
            return (K.blend_dev(g, None, ctx.alpha, 1) if ctx.needs_input_grad[0] else None,
                    K.blend_dev(g, None, ctx.alpha, 2) if ctx.needs_input_grad[1] else None, None)
        return (K.axpby(g, None, 1.0 - ctx.alpha) if ctx.needs_input_grad[0] else None,
                K.axpby(g, None, ctx.alpha) if ctx.needs_input_grad[1] else None, None)


def blend(a, b, alpha):
    return _Blend.apply(a, b, alpha)


class _MinibatchStd(Function):
    @staticmethod
    def forward(ctx, x):
        y, ws = K.minibatch_std_fwd(x)
        ctx.save_for_backward(x, ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ws = ctx.saved_tensors
        return K.minibatch_std_bwd(_c(dy), x, ws)


def minibatch_std(x):
    """PGGAN/model_nvidia.py:20-29"""
    return _MinibatchStd.apply(_c(x))


class _ConcatC(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[-1]
        return K.concat_channels(_c(a), _c(b))

    @staticmethod
    def backward(ctx, g):
        return K.split_channels(_c(g), ctx.ca)


def concat_channels(a, b):
    """tf.concat([a, b], axis=3)"""
    return _ConcatC.apply(a, b)


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, keep, rng_state):
        y, mask = K.dropout_fwd(_c(x), keep, rng_state)
        ctx.save_for_backward(mask)
        ctx.keep = keep
        return y

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return K.dropout_bwd(_c(g), mask, ctx.keep), None, None


def dropout(x, keep_prob, rng_state):
    """tf.nn.dropout(x, keep_prob)"""
    return _Dropout.apply(x, keep_prob, rng_state)


class _L1(Function):
    @staticmethod
    def forward(ctx, a, b):
        loss, dl32 = K.l1_loss(_c(a), _c(b))
        ctx.save_for_backward(dl32)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl32,) = ctx.saved_tensors
        return K.loss_grad_scale(dl32, _c(g.to(torch.float32)).reshape(1)), None


def l1_loss(a, b):
    """tf.reduce_mean(tf.abs(a - b)); differentiable in a (the target b is data)"""
    return _L1.apply(a, b)
