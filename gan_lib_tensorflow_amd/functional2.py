"""Twice-differentiable operators: the critic of the ACGAN configuration under the WGAN-GP term.

ACGAN/train.py:99-107 differentiates the critic's INPUT gradient with respect to the critic's weights
(`tf.gradients(D(interpolates), [interpolates])` inside the loss).  TensorFlow gets the second-order graph from its op
registry; here every operator of that critic (ACGAN/model.py:49-90) is a `torch.autograd.Function` whose backward is
written in terms of OTHER Functions of this module, so autograd can differentiate the backward pass again:

    conv   : ConvF (fprop)  <->  ConvD (input gradient)  <->  ConvW (filter gradient)     -- closed under differentiation,
             all three on the same MFMA kernels as the first-order path (gank_conv2d_fprop / _dgrad / _wgrad);
    dense  : LinF / LinD / LinW the same way on gank_linear_fwd / _bwd;
    lrelu  : LRelu -> LReluB (mask multiply; linear in dy, zero second derivative in x);
    pooling: Pool2 <-> Unpool2, SumHW <-> BcastHW (linear, mutually adjoint);
    batch norm (train mode): BNF -> BNB -> gank_bn_bwd_bwd (its second derivative is a kernel of its own).

Weight gradients are RETURNED (autograd accumulates them into `p.grad`, which the trainer points at the flat gradient
buffer): a filter receives first- and second-order contributions in one backward pass.  Activations bf16 NHWC, weights fp32.
"""
import torch
from torch.autograd import Function

from . import kernels as K

BF16 = K.BF16


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _geom(W):
    if W.dim() == 2:
        return 1, W.shape[0], W.shape[1]
    return W.shape[0], W.shape[2], W.shape[3]


# MFMA operand layouts of a weight, prepared once per update and reused by every pass over the critic in it (real, fake,
# interpolates, and the second-order terms) inside a `with one_update():` block,
# so a captured update holds exactly one preparation launch per weight and layout.  Keyed by the parameter tensor (leaves
# only: a weight that is itself a graph node -- the `g` of a second-order term -- is prepared where it is used).
_prep_cache = {}
_prep_cache_on = [False]


class one_update:
    """with one_update(): ...one update's passes...  -- weights do not change inside (the optimiser runs after it); outside
    such a block every call prepares its own operands (direct model calls, tests that rewrite the weights in between)"""

    def __enter__(self):
        _prep_cache.clear()
        self.prev, _prep_cache_on[0] = _prep_cache_on[0], True
        return self

    def __exit__(self, *exc):
        _prep_cache_on[0] = self.prev
        _prep_cache.clear()
        return False


def _prepared(W, want_f):
    k, cin, cout = _geom(W)
    if not W.is_leaf or not _prep_cache_on[0]:
        return K.prep_weights(_c(W.detach().to(torch.float32)).view(k, k, cin, cout), want_f, not want_f)[0 if want_f else 1]
    key = (id(W), want_f)
    hit = _prep_cache.get(key)
    if hit is None or hit[0] is not W:
        op = K.prep_weights(_c(W.detach().to(torch.float32)).view(k, k, cin, cout), want_f, not want_f)[0 if want_f else 1]
        _prep_cache[key] = hit = (W, op)
    return hit[1]


def prepare_batched(ws):
    """Inside `with one_update():` -- both operand layouts of every 4-D leaf weight of `ws` in ONE launch
    (kernels.prep_weights_batched, the layouts of K.prep_weights bit for bit) instead of one launch per weight and layout on
    first use: a critic update held 22 such launches.  The buffers persist on the tensors (`w._prep`, rewritten in place by
    later calls: captured graphs keep reading the same addresses); the cache entries live until the block ends."""
    if not _prep_cache_on[0]:
        return
    ws = [w for w in ws if w.is_leaf and w.dim() == 4 and w.is_cuda and w.dtype == torch.float32]
    if not ws:
        return
    K.prep_weights_batched(ws, want_d=True)
    for w in ws:
        wf, wd = w._prep
        _prep_cache[(id(w), True)] = (w, wf)
        _prep_cache[(id(w), False)] = (w, wd)


def _wf(W):
    return _prepared(W, True)


def _wd(W):
    return _prepared(W, False)


def _direct(p):
    """In a backward pass that is not itself differentiated, a parameter whose `.grad` already exists receives its gradient by
    in-place addition (AccumulateGrad).  The filter / bias gradient kernels ACCUMULATE into their target, so they can write
    there themselves: no zero fill of a temporary, no add launch -- 2 of the ~6 launches a small layer's backward costs.
    -> the buffer to accumulate into, or None (differentiated pass, non-leaf weight, no gradient buffer yet)."""
    if torch.is_grad_enabled() or p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


# A backward pass that is wanted for its INPUT gradient only -- tf.gradients(D(x_hat), [x_hat]) inside the WGAN-GP term
# (ACGAN/train.py:101-103).  autograd calls every node's backward with needs_input_grad as recorded at forward time (the
# weights DO require gradients), so without this hint every layer of that pass also computes -- differentiably -- a filter
# and bias gradient that nothing reads: 10 filter-gradient launches, 10 column sums and 20 zero fills per critic update.
_dx_only = [False]


class input_gradient_only:
    """with input_gradient_only(): torch.autograd.grad(outputs, [activation], create_graph=True)"""

    def __enter__(self):
        self.prev, _dx_only[0] = _dx_only[0], True
        return self

    def __exit__(self, *exc):
        _dx_only[0] = self.prev
        return False


class ConvF(Function):
    """y = conv2d_SAME(x, W) + b   (conv2d.py:180-187,212-216)"""

    @staticmethod
    def forward(ctx, x, W, bias):
        k, cin, cout = _geom(W)
        ctx.save_for_backward(x, W)
        ctx.has_bias = bias is not None
        ctx.bias = bias                      # the parameter itself (for its .grad buffer), not a saved value
        n, h, w, _ = x.shape
        return K.conv2d_fprop(_c(x), _wf(W), bias.detach() if bias is not None else None, (h, w), cout, k)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _c(dy)
        dx = ConvD.apply(dy, W) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1] and not _dx_only[0]
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not _dx_only[0]
        if want_w and want_b:
            tw, tb = _direct(W), _direct(ctx.bias)
            if tw is not None and tb is not None:        # both accumulate in place: the bias gradient rides on the filter-gradient launch
                k, cin, cout = _geom(W)
                n, h, w, _ = x.shape
                K.conv2d_wgrad(_c(x), dy, tw.view(k, k, cin, cout), (h, w), k, 0, 1.0, dbias=tb)
                return dx, None, None
        dW = _conv_wgrad(x, dy, W) if want_w else None
        db = _colsum(dy, ctx.bias) if want_b else None
        return dx, dW, db


class ConvD(Function):
    """dx = conv2d_SAME(dy, flip(W)^T): the adjoint of ConvF in x (Conv2DBackpropInput)"""

    @staticmethod
    def forward(ctx, dy, W):
        k, cin, cout = _geom(W)
        ctx.save_for_backward(dy, W)
        n, h, w, _ = dy.shape
        return K.conv2d_dgrad(_c(dy), _wd(W), (h, w), cin, k)

    @staticmethod
    def backward(ctx, g):
        dy, W = ctx.saved_tensors
        g = _c(g)
        d_dy = ConvF.apply(g, W, None) if ctx.needs_input_grad[0] else None
        dW = _conv_wgrad(g, dy, W) if ctx.needs_input_grad[1] else None
        return d_dy, dW


def _conv_wgrad(x, dy, W):
    tgt = _direct(W)
    if tgt is None:
        return ConvW.apply(x, dy, W.shape)
    k, cin, cout = _geom(W)
    n, h, w, _ = x.shape
    K.conv2d_wgrad(_c(x), _c(dy), tgt.view(k, k, cin, cout), (h, w), k)
    return None


def _colsum(dy, bias):
    tgt = _direct(bias)
    if tgt is None:
        return ColSum.apply(dy)
    K.colsum(_c(dy), tgt, 1.0)
    return None


class ConvW(Function):
    """dW[tap, ci, co] = sum_pixels x[p + tap, ci] * dy[p, co]   (Conv2DBackpropFilter), fp32"""

    @staticmethod
    def forward(ctx, x, dy, wshape):
        ctx.save_for_backward(x, dy)
        ctx.wshape = tuple(wshape)
        k = 1 if len(wshape) == 2 else wshape[0]
        cin, cout = wshape[-2], wshape[-1]
        dw = torch.zeros((k, k, cin, cout), dtype=torch.float32, device=x.device)
        n, h, w, _ = x.shape
        K.conv2d_wgrad(_c(x), _c(dy), dw, (h, w), k)
        return dw.view(ctx.wshape)

    @staticmethod
    def backward(ctx, g):
        x, dy = ctx.saved_tensors
        g = _c(g).view(ctx.wshape)
        dx = ConvD.apply(dy, g) if ctx.needs_input_grad[0] else None
        d_dy = ConvF.apply(x, g, None) if ctx.needs_input_grad[1] else None
        return dx, d_dy, None


class ColSum(Function):
    """bias gradient: sum over pixels (tf.nn.bias_add gradient)"""

    @staticmethod
    def forward(ctx, dy):
        out = torch.zeros(dy.shape[-1], dtype=torch.float32, device=dy.device)
        return K.colsum(_c(dy), out, 1.0)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("second derivative through a bias gradient: the gradient penalty only uses the input gradient")


class LinF(Function):
    """y = x W + b on the fp32 weight (linear.py:161-180)"""

    @staticmethod
    def forward(ctx, x, W, bias):
        ctx.save_for_backward(x, W)
        ctx.has_bias = bias is not None
        ctx.bias = bias
        return K.linear_fwd(_c(x), _c(W.detach().to(torch.float32)), bias.detach() if bias is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _c(dy)
        dx = LinD.apply(dy, W) if ctx.needs_input_grad[0] else None
        dW = _lin_wgrad(x, dy, W) if (ctx.needs_input_grad[1] and not _dx_only[0]) else None
        db = _colsum(dy, ctx.bias) if (ctx.has_bias and ctx.needs_input_grad[2] and not _dx_only[0]) else None
        return dx, dW, db


class LinD(Function):
    """dx = dy W^T"""

    @staticmethod
    def forward(ctx, dy, W):
        ctx.save_for_backward(dy, W)
        return K.linear_bwd(_c(dy), None, _c(W.detach().to(torch.float32)), True)

    @staticmethod
    def backward(ctx, g):
        dy, W = ctx.saved_tensors
        g = _c(g)
        d_dy = LinF.apply(g, W, None) if ctx.needs_input_grad[0] else None
        dW = _lin_wgrad(g, dy, W) if ctx.needs_input_grad[1] else None
        return d_dy, dW


def _lin_wgrad(x, dy, W):
    tgt = _direct(W)
    if tgt is None or tgt.dim() != 2:
        return LinW.apply(x, dy)
    K.linear_bwd(_c(dy), _c(x), None, False, tgt, None)
    return None


class LinW(Function):
    """dW = x^T dy, fp32 [K, C]"""

    @staticmethod
    def forward(ctx, x, dy):
        ctx.save_for_backward(x, dy)
        dw = torch.zeros((x.shape[1], dy.shape[1]), dtype=torch.float32, device=x.device)
        K.linear_bwd(_c(dy), _c(x), None, False, dw, None)
        return dw

    @staticmethod
    def backward(ctx, g):
        x, dy = ctx.saved_tensors
        dx = LinD.apply(dy, g) if ctx.needs_input_grad[0] else None
        d_dy = LinF.apply(x, g, None) if ctx.needs_input_grad[1] else None
        return dx, d_dy


class LRelu(Function):
    """tf.maximum(x, leak * x)   (resnet_block.py:24-29); leak = 0 is relu"""

    @staticmethod
    def forward(ctx, x, leak):
        ctx.save_for_backward(x)
        ctx.leak = leak
        return K.relu_fwd(_c(x), leak)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return LReluB.apply(dy, x, ctx.leak), None


class LReluB(Function):
    """dx = dy * (x > 0 ? 1 : leak): linear in dy, piecewise constant in x"""

    @staticmethod
    def forward(ctx, dy, x, leak):
        ctx.save_for_backward(x)
        ctx.leak = leak
        return K.relu_bwd(_c(dy), x, leak)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return LReluB.apply(g, x, ctx.leak), None, None


class Pool2(Function):
    """y = scale * (sum of each 2x2 block): the reference's add_n of four strided slices / 4 (resnet_block.py:62-64)"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return K.pool2x2(_c(x), scale)

    @staticmethod
    def backward(ctx, dy):
        return Unpool2.apply(dy, ctx.scale), None


class Unpool2(Function):
    @staticmethod
    def forward(ctx, g, scale):
        ctx.scale = scale
        return K.unpool2x2_add(_c(g), None, scale)

    @staticmethod
    def backward(ctx, g2):
        return Pool2.apply(g2, ctx.scale), None


class SumHW(Function):
    """y[n, c] = scale * sum over (h, w): tf.reduce_mean(axis=[1, 2]) with scale = 1 / (H W)   (model.py:72)"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.hw, ctx.scale = (x.shape[1], x.shape[2]), scale
        return K.sum_hw(_c(x), scale)

    @staticmethod
    def backward(ctx, dy):
        return BcastHW.apply(dy, ctx.hw, ctx.scale), None


class BcastHW(Function):
    @staticmethod
    def forward(ctx, g, hw, scale):
        ctx.scale = scale
        return K.bcast_hw(_c(g), hw, scale)

    @staticmethod
    def backward(ctx, g2):
        return SumHW.apply(g2, ctx.scale), None, None


_zero_labels = {}
_bnb_scratch = {}      # (device, C) -> fp32 [2, 1, C] rows the batch-norm backward may accumulate into when nobody wants the table gradients


def _zl(x):
    key = (x.shape[0], str(x.device))
    if key not in _zero_labels:
        _zero_labels[key] = torch.zeros(x.shape[0], dtype=torch.int32, device=x.device)
    return _zero_labels[key]


class BNF(Function):
    """train-mode batch norm over (N, H, W): y = gamma (x - mean) * invstd + beta; second output = [mean, invstd] ([2, C], not
    differentiable: feeds the moving-statistics update)"""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = _c(x)
        y, stats = K.cbn_fwd(x, _zl(x), gamma.detach().view(1, -1), beta.detach().view(1, -1), 1, False)
        stats = stats.view(2, -1)
        ctx.save_for_backward(x, gamma, stats)
        ctx.beta = beta                      # the parameter itself (for its .grad buffer)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # the statistics output has no gradient: it must not arrive as a zero-filled tensor
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        if dy is None:
            return None, None, None
        x, gamma, stats = ctx.saved_tensors
        tg, tb = _direct(gamma), _direct(ctx.beta)
        if tg is not None and tb is not None:        # not differentiated again: the table-gradient kernel adds into .grad itself
            c = x.shape[-1]
            dx = K.cbn_bwd(_c(dy), x, x, _zl(x), gamma.detach().view(1, -1), stats.view(1, 2, c), tg.view(1, c), tb.view(1, c), 1, False)
            return dx, None, None
        # differentiated again (the gradient-penalty term): when this pass wants the input gradient only -- autograd.grad of the
        # critic output w.r.t. the interpolates, ACGAN/train.py:101-103 -- the table gradients are neither zero-filled nor kept
        want = (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and not _dx_only[0]
        dx, dgamma, dbeta = BNB.apply(dy, x, gamma, stats, want)
        return dx, (dgamma if want else None), (dbeta if want else None)


class BNB(Function):
    """first backward of BNF: (dx, dgamma, dbeta)"""

    @staticmethod
    def forward(ctx, dy, x, gamma, stats, want_tables=True):
        dy = _c(dy)
        c = x.shape[-1]
        if want_tables:
            dgamma = torch.zeros((1, c), dtype=torch.float32, device=x.device)
            dbeta = torch.zeros((1, c), dtype=torch.float32, device=x.device)
        else:
            # the kernel accumulates its table gradients somewhere: a persistent scratch row that nobody reads (no fill launches)
            key = (str(x.device), c)
            if key not in _bnb_scratch:
                _bnb_scratch[key] = torch.zeros((2, 1, c), dtype=torch.float32, device=x.device)
            dgamma, dbeta = _bnb_scratch[key][0], _bnb_scratch[key][1]
        dx = K.cbn_bwd(dy, x, x, _zl(x), gamma.detach().view(1, -1), stats.view(1, 2, c), dgamma, dbeta, 1, False)
        ctx.save_for_backward(dy, x, gamma, stats)
        ctx.set_materialize_grads(False)
        if not want_tables:
            return dx, None, None
        return dx, dgamma.view(gamma.shape), dbeta.view(gamma.shape)

    @staticmethod
    def backward(ctx, g_dx, g_dgamma, g_dbeta):
        dy, x, gamma, stats = ctx.saved_tensors
        if g_dgamma is not None or g_dbeta is not None:
            raise NotImplementedError("second derivative through dgamma / dbeta: the gradient penalty only uses the input gradient")
        if g_dx is None:
            return None, None, None, None, None
        tg = _direct(gamma)
        if tg is not None:                       # the last pass: the kernel adds gamma's second-order term into .grad itself
            gI, ggO = K.bn_bwd_bwd(_c(g_dx), dy, x, _c(gamma.detach().view(-1)), _c(stats.view(-1)), tg.view(-1))
            return ggO, gI, None, None, None
        gG = torch.zeros(gamma.numel(), dtype=torch.float32, device=x.device)
        gI, ggO = K.bn_bwd_bwd(_c(g_dx), dy, x, _c(gamma.detach().view(-1)), _c(stats.view(-1)), gG)
        return ggO, gI, gG.view(gamma.shape), None, None


class GPLoss(Function):
    """lambda * mean_n (||g_n||_2 - 1)^2 with the reference's sqrt(sum + 1e-10)   (ACGAN/train.py:104-106)"""

    @staticmethod
    def forward(ctx, grad, lam):
        loss, dg = K.gp_loss(_c(grad), lam)
        ctx.save_for_backward(dg)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dg32,) = ctx.saved_tensors
        return K.loss_grad_scale(dg32, _c(g.to(torch.float32)).reshape(1)), None       # fp32 product, one rounding to bf16


# ---- functional forms -------------------------------------------------------------------------------------------
class AddF(Function):
    """a + b (one launch); linear, so its backward is differentiable as it stands"""

    @staticmethod
    def forward(ctx, a, b):
        return K.add(_c(a), _c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


class Fork(Function):
    """Explicit activation fan-out (an activation read by a block's shortcut and by its main path, the pooled features read by
    both heads): two aliases forward; backward ONE add launch of this library instead of autograd's own accumulation (an
    at::native add), again through a Function so that the gradient-penalty pass can be differentiated through it."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        return AddF.apply(ga, gb)


def fork(x):
    if not x.requires_grad:
        return x, x
    return Fork.apply(x)


def add(a, b):
    return AddF.apply(a, b)


def conv2d(x, W, bias=None):
    return ConvF.apply(x, W, bias)


def linear(x, W, bias=None):
    return LinF.apply(x, W, bias)


def lrelu(x, leak=0.2):
    return LRelu.apply(x, leak)


def meanpool2x2(x):
    return Pool2.apply(x, 0.25)


def mean_hw(x):
    return SumHW.apply(x, 1.0 / (x.shape[1] * x.shape[2]))


def batch_norm_train(x, gamma, beta):
    """-> (y, stats [2, C])"""
    return BNF.apply(x, gamma, beta)


def gradient_penalty(grad, lam=10.0):
    return GPLoss.apply(grad, lam)
