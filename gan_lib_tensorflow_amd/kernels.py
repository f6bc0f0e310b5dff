"""Tensor-level wrappers over the C ABI (include/gank.h).  torch is used for device memory and
streams only: every function checks its operands, then enqueues HIP kernels from libgank.so on the
current torch stream.  Nothing here has a CPU path."""
import ctypes as C

import torch

from . import _lib
from ._lib import IN_UPSAMPLE2X, IN_RELU, OUT_TANH, DY_UPSAMPLE2X, RES_UPSAMPLE2X, STATS_PREZEROED, OUT_POOLSUM2X, SnDesc, PrepDesc, WgradItem, LabelDenseDesc, Res8Head, SlabJob  # noqa: F401

# BF16 = the 16-bit activation dtype of this process: torch.bfloat16, or torch.float16 under GANK_DTYPE=fp16 (libgank_f16.so)
BF16, F32, I32 = getattr(torch, _lib.ACT_DTYPE_NAME), torch.float32, torch.int32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"gank: {name} must live on the GPU (no CPU path exists)")
    if not t.is_contiguous():
        raise RuntimeError(f"gank: {name} must be contiguous, got strides {t.stride()} for {tuple(t.shape)}")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"gank: {name} must be {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _roundup(a, b):
    return (a + b - 1) // b * b


def lib():
    return _lib.load()


# ------------------------------------------------------------------ conv
def prep_weights(w, want_f=True, want_d=False):
    """w fp32 [k,k,Cin,Cout] -> (wf bf16 [CoutPad,Kpad] | None, wd bf16 [CinPad,Kpad'] | None)"""
    k, _, cin, cout = w.shape
    taps = k * k
    wf = torch.empty((_roundup(cout, 32), _roundup(taps * cin, 64)), dtype=BF16, device=w.device) if want_f else None
    wd = torch.empty((_roundup(cin, 32), _roundup(taps * cout, 64)), dtype=BF16, device=w.device) if want_d else None
    _lib.check(lib().gank_conv2d_prep_weights(_p(w, F32, "w"), _p(wf), _p(wd), k, cin, cout, _stream()), "prep_weights")
    return wf, wd


_PREP_ATTR = ("_prep", "_prep_up", "_prep_pool", None, "_prep_res", "_prep_cpres", "_prep_feat")     # kind 3 is retired; 6: below


def _prep_plan(ws, want_d=True, kinds=None, sources=None):
    """Descriptor table for the batched operand preparation of `ws` (see prep_weights_batched).  sources[i]: the tensor
    whose VALUES entry i reads (default: ws[i] itself) -- the fused spectral-norm launch reads the master weight and divides
    by sigma on the fly, while the operands belong to the normalised tensor.  -> (table, todo indices, outs)"""
    kinds = list(kinds) if kinds is not None else [0] * len(ws)
    todo = [i for i, kd in enumerate(kinds) if kd is not None]
    table = (PrepDesc * max(len(todo), 1))()
    outs = []
    for slot, i in enumerate(todo):
        w, kind = ws[i], kinds[i]
        if w.dim() == 2:
            k, cin, cout = 1, w.shape[0], w.shape[1]
        else:
            k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
        taps = k * k
        dev = w.device
        if kind == 0:
            old = getattr(w, "_prep", None)
            if old is not None and old[0] is not None and (old[1] is not None or not want_d):
                wf, wd = old
            else:
                wf = torch.empty((_roundup(cout, 32), _roundup(taps * cin, 64)), dtype=BF16, device=dev)
                wd = torch.empty((_roundup(cin, 32), _roundup(taps * cout, 64)), dtype=BF16, device=dev) if want_d else None
        elif kind == 1:
            assert k == 3
            wf, wd = getattr(w, "_prep_up", None) or (torch.empty((4, _roundup(cout, 32), 4 * cin), dtype=BF16, device=dev),
                                                     torch.empty((_roundup(cin, 32), _roundup(16 * cout, 64)), dtype=BF16, device=dev))
        elif kind == 2:
            assert k == 3
            wf, wd = getattr(w, "_prep_pool", None) or (torch.empty((_roundup(cout, 32), _roundup(16 * cin, 64)), dtype=BF16, device=dev),
                                                       torch.empty((4, _roundup(cin, 32), 4 * cout), dtype=BF16, device=dev))
        elif kind == 4:
            assert k == 3 and cin % 32 == 0 and cout % 32 == 0
            wf, wd = getattr(w, "_prep_res", None) or (torch.empty(taps * cin * cout, dtype=BF16, device=dev),
                                                      torch.empty(taps * cin * cout, dtype=BF16, device=dev) if want_d else None)
        elif kind == 5:
            assert k == 3 and cin % 64 == 0 and cout % 32 == 0
            wf, wd = getattr(w, "_prep_cpres", None) or (torch.empty(16 * cin * cout, dtype=BF16, device=dev),
                                                        torch.empty(16 * cin * cout, dtype=BF16, device=dev))
        elif kind == 6:
            # "rfrag" operands (kind 4) of the FIRST HALF of the input channels: the feature half of a conv whose other input
            # channels are spatially constant and factored out (label_conv3x3_table) -> `w._prep_feat`
            assert k == 3 and cin % 64 == 0 and cout % 32 == 0
            half = cin // 2
            wf, wd = getattr(w, "_prep_feat", None) or (torch.empty(taps * half * cout, dtype=BF16, device=dev),
                                                       torch.empty(taps * half * cout, dtype=BF16, device=dev) if want_d else None)
        else:
            raise ValueError(f"unknown preparation kind {kind}")
        src = w if sources is None else sources[i]
        d = table[slot]
        d.w, d.wf, d.wd = _p(src.detach(), F32, "w").value, wf.data_ptr(), (wd.data_ptr() if wd is not None else None)
        d.ksize, d.Cin, d.Cout, d.kind = k, cin, cout, kind
        if kind == 6:
            d.Cin, d.kind, d.cin_pitch = cin // 2, 4, cin
        outs.append((wf, wd))
    return table, todo, outs


def _prep_attach(ws, kinds, todo, outs):
    for i, o in zip(todo, outs):
        setattr(ws[i], _PREP_ATTR[kinds[i]], o)


def prep_weights_batched(ws, want_d=True, kinds=None):
    """One launch for a list of fp32 weights ([k,k,Cin,Cout] or [Cin,Cout]).  kinds[i]: 0 = plain conv/linear ->
    `w._prep = (wf, wd)`; 1 = UpsampleConv 3x3 -> `w._prep_up = (wph, wd4)`; 2 = ConvMeanPool 3x3 ->
    `w._prep_pool = (wp4, wphd)`; 4 = "rfrag" operands of the resident kernels -> `w._prep_res = (rf, rd)`;
    5 = ConvMeanPool 3x3 operands of the resident kernels -> `w._prep_cpres = (rf, rd)`;
    None = skip (the layer prepares nothing).  The conv wrappers pick the attributes up
    and skip their own per-layer preparation.  Buffers persist on the tensor and are rewritten IN PLACE on later
    calls: captured graphs keep reading the same addresses."""
    kinds = list(kinds) if kinds is not None else [0] * len(ws)
    table, todo, outs = _prep_plan(ws, want_d, kinds)
    if todo:
        _lib.check(lib().gank_conv2d_prep_weights_batched(table, len(todo), _stream()), "prep_weights_batched")
    _prep_attach(ws, kinds, todo, outs)
    return outs


class stats_arena:
    """One zero fill for the statistics sums of every conv of a pass: inside the context, conv2d_fprop / upconv3x3_fprop
    take their `sums` buffers from one pre-cleared allocation (GANK_STATS_PREZEROED) instead of each launching a fill.
    `floats`: capacity; a request that does not fit falls back to its own buffer and fill."""
    current = None

    def __init__(self, floats, device, buf=None):
        # buf: a cleared buffer of `floats` floats the caller already has (generator_feed)
        if buf is not None:
            assert buf.numel() >= int(floats) and buf.dtype == F32
            self.buf = buf
        else:
            self.buf = zeros_f32(int(floats), device) if torch.device(device).type == 'cuda' else torch.zeros(int(floats), dtype=F32, device=device)
        self.used = 0

    def __enter__(self):
        self.prev, stats_arena.current = stats_arena.current, self
        return self

    def __exit__(self, *exc):
        stats_arena.current = self.prev
        return False

    @staticmethod
    def take(shape, device):
        """-> (buffer, prezeroed)"""
        n = 1
        for d in shape:
            n *= d
        ar = stats_arena.current
        if ar is not None and ar.buf.device == device and ar.used + n <= ar.buf.numel():
            out = ar.buf[ar.used:ar.used + n].view(shape)
            ar.used += (n + 63) // 64 * 64
            return out, True
        return torch.empty(shape, dtype=F32, device=device), False


class ConvStats:
    """Batch-norm statistics a conv epilogue accumulated for the layer that consumes its output (gank_conv2d_fprop_stats):
    sums [groups][STAT_SLOTS][2][C]: partial sums of (y - shift) and its square per tower; `shift` = the conv's bias (or None)."""
    __slots__ = ("sums", "shift", "groups")

    def __init__(self, sums, shift, groups):
        self.sums, self.shift, self.groups = sums, shift, groups


def conv2d_fprop(x, wf, bias, out_hw, cout, ksize, flags=0, scale=1.0, residual=None, relu_ref=None, stats_groups=0):
    """stats_groups > 0: also returns a ConvStats (or None when the kernel that ran does not produce them) -> (y, stats)"""
    n, cin = x.shape[0], x.shape[3]
    h, w = out_hw
    y = torch.empty((n, h, w, cout), dtype=BF16, device=x.device)
    if stats_groups:
        sums, pre = stats_arena.take((stats_groups, _lib.STAT_SLOTS, 2, cout), x.device)
        produced = C.c_int(0)
        _lib.check(lib().gank_conv2d_fprop_stats(_p(x, BF16, "x"), _p(wf, BF16, "wf"), _p(bias, F32, "bias"),
                                                 _p(residual, BF16, "residual"), _p(relu_ref, BF16, "relu_ref"), _p(y),
                                                 n, h, w, cin, cout, ksize, flags | (STATS_PREZEROED if pre else 0), scale, _p(sums), stats_groups, C.byref(produced),
                                                 _stream()), "conv2d_fprop_stats")
        return y, (ConvStats(sums, bias, stats_groups) if produced.value else None)
    _lib.check(lib().gank_conv2d_fprop(_p(x, BF16, "x"), _p(wf, BF16, "wf"), _p(bias, F32, "bias"),
                                       _p(residual, BF16, "residual"), _p(relu_ref, BF16, "relu_ref"), _p(y),
                                       n, h, w, cin, cout, ksize, flags, scale, _stream()), "conv2d_fprop")
    return y


def meanpool_conv1x1_fprop(x, wf, bias, cout, keep_pooled=True):
    """conv1x1(mean_pool2x2(x)) + bias on a 3-channel image, the pool inside the gather -> (y [N,H/2,W/2,Cout], pooled [N,H/2,W/2,3] | None)"""
    n, h2, w2, cin = x.shape
    h, w = h2 // 2, w2 // 2
    y = torch.empty((n, h, w, cout), dtype=BF16, device=x.device)
    pooled = torch.empty((n, h, w, cin), dtype=BF16, device=x.device) if keep_pooled else None
    _lib.check(lib().gank_meanpool_conv1x1_fprop(_p(x, BF16, "x"), _p(wf, BF16, "wf"), _p(bias, F32, "bias"), _p(y), _p(pooled),
                                                 n, h, w, cin, cout, _stream()), "meanpool_conv1x1_fprop")
    return y, pooled


def image_conv_pair_fprop(x, wf1, bias1, cout1, wfs, biass, couts, keep_pooled=True):
    """conv2d_fprop(x, wf1, bias1, 3x3) and meanpool_conv1x1_fprop(x, wfs, biass) in one launch (gank_image_conv_pair_fprop)
    -> (y1 [N,H,W,cout1], ys [N,H/2,W/2,couts], pooled image | None)"""
    n, h, w, cin = x.shape
    assert cin == 3 and h % 2 == 0 and w % 2 == 0
    y1 = torch.empty((n, h, w, cout1), dtype=BF16, device=x.device)
    ys = torch.empty((n, h // 2, w // 2, couts), dtype=BF16, device=x.device)
    pooled = torch.empty((n, h // 2, w // 2, cin), dtype=BF16, device=x.device) if keep_pooled else None
    _lib.check(lib().gank_image_conv_pair_fprop(_p(x, BF16, "x"), _p(wf1, BF16, "wf1"), _p(bias1, F32, "bias1"), _p(y1), _p(wfs, BF16, "wfs"),
                                                _p(biass, F32, "biass"), _p(ys), _p(pooled), n, h, w, cout1, couts, _stream()), "image_conv_pair_fprop")
    return y1, ys, pooled


def conv2d_dgrad(dy, wd, out_hw, cin, ksize, flags=0, scale=1.0, residual=None, relu_ref=None):
    n, cout = dy.shape[0], dy.shape[3]
    h, w = out_hw
    dx = torch.empty((n, h, w, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_conv2d_dgrad(_p(dy, BF16, "dy"), _p(wd, BF16, "wd"), _p(residual, BF16, "residual"),
                                       _p(relu_ref, BF16, "relu_ref"), _p(dx), n, h, w, cin, cout, ksize, flags,
                                       scale, _stream()), "conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, dw, hw, ksize, flags=0, scale=1.0, dbias=None, slab_jobs=None):
    """ACCUMULATES into dw fp32 [k,k,Cin,Cout] (and the bias gradient into dbias fp32 [Cout] when given).
    slab_jobs: a list -- small filters on the split-K kernels that add their partial tiles with fp32 atomics write per-split
    slabs instead, and the job that sums them into dw is appended: the caller owes a sum_slabs(list) (dbias: as before)."""
    n, cin, cout = x.shape[0], x.shape[3], dy.shape[3]
    assert dw.shape[-2] == cin and dw.shape[-1] == cout, (dw.shape, cin, cout)
    if slab_jobs is not None:
        slab_elems = int(lib().gank_conv2d_wgrad_slab_elems(n, hw[0], hw[1], cin, cout, ksize, flags))
        if slab_elems > 0:
            ws = torch.empty(slab_elems, dtype=F32, device=x.device)
            job = (SlabJob * 1)()
            _lib.check(lib().gank_conv2d_wgrad_slabs(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"),
                                                     n, hw[0], hw[1], cin, cout, ksize, flags, scale, _p(ws), slab_elems, job, _stream()),
                       "conv2d_wgrad_slabs")
            if job[0].nslabs > 0:
                slab_jobs.append((job, 1, ws))
            return dw
    ws_elems = lib().gank_conv2d_wgrad_ws_elems(n, hw[0], hw[1], cin, cout, ksize, flags)
    ws = torch.empty(ws_elems, dtype=F32, device=x.device) if ws_elems > 0 else None
    _lib.check(lib().gank_conv2d_wgrad(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"),
                                       _p(ws), ws_elems, n, hw[0], hw[1], cin, cout, ksize, flags, scale, _stream()),
               "conv2d_wgrad")
    return dw


def conv1x1_wgrad_dgrad_ok(n, hw, cin, cout, wd):
    return (cin % 64 == 0 and cout % 16 == 0 and cout <= 128 and (n * hw[0] * hw[1]) % 32 == 0 and wd is not None and wd.dim() == 2
            and wd.shape[0] >= cin and wd.shape[1] >= cout and wd.shape[1] % 8 == 0)


def conv1x1_wgrad_dgrad(x, dy, dw, wd, dbias=None):
    """gank_conv1x1_wgrad_dgrad: ACCUMULATES the 1x1 filter gradient into dw [.., Cin, Cout] (and dbias) and returns the input gradient
    dx [N,H,W,Cin] = dy wd^T from extra workgroups of the same launch"""
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    assert dw.shape[-2] == cin and dw.shape[-1] == cout and tuple(dy.shape[:3]) == (n, h, w)
    dx = torch.empty((n, h, w, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_conv1x1_wgrad_dgrad(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"), _p(wd, BF16, "wd"),
                                              wd.shape[1], _p(dx), n, h, w, cin, cout, _stream()), "conv1x1_wgrad_dgrad")
    return dx


def conv2d_wgrad_rows_ok(n, hw, cin, cout, ksize=3, flags=0):
    """the layer has the deferred slab form conv2d_wgrad_rows needs"""
    return int(lib().gank_conv2d_wgrad_slab_splits(n, hw[0], hw[1], cin, cout, ksize, flags)) > 0


def conv2d_wgrad_rows(x, dy, dw_full, hw, ksize, flags, slab_jobs, dbias=None, tap_sums=None):
    """filter gradient of the FIRST x.shape[3] input channels of the wider filter dw_full [k,k,Cin_total,Cout], accumulated into those
    rows by a slab job appended to slab_jobs (gank_conv2d_wgrad_slabs_rows; the caller owes a sum_slabs(list)).  tap_sums = (lists,
    n_labels): the launch also computes the per-label tap sums of dy (extra workgroups) -> their workspace, for
    label_conv3x3_bwd(sums=...)"""
    n, cin, cout = x.shape[0], x.shape[3], dy.shape[3]
    assert dw_full.dim() == 4 and dw_full.shape[3] == cout and dw_full.shape[2] >= cin and slab_jobs is not None
    slab_elems = int(lib().gank_conv2d_wgrad_slab_elems(n, hw[0], hw[1], cin, cout, ksize, flags))
    ws = torch.empty(max(slab_elems, 1), dtype=F32, device=x.device)
    job = (SlabJob * 1)()
    sums = None
    if tap_sums is not None:
        lists, v = tap_sums
        sums = torch.empty(int(lib().gank_label_conv3x3_bwd_ws_floats(n, cout)), dtype=F32, device=x.device)
        _lib.check(lib().gank_conv2d_wgrad_slabs_rows_tap_sums(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw_full, F32, "dw_full"), _p(dbias, F32, "dbias"),
                                                               n, hw[0], hw[1], cin, dw_full.shape[2], cout, ksize, flags, 1.0, _p(ws), slab_elems, job,
                                                               _p(lists, I32, "lists"), int(v), _p(sums), _stream()), "conv2d_wgrad_slabs_rows_tap_sums")
    else:
        _lib.check(lib().gank_conv2d_wgrad_slabs_rows(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw_full, F32, "dw_full"), _p(dbias, F32, "dbias"),
                                                      n, hw[0], hw[1], cin, dw_full.shape[2], cout, ksize, flags, 1.0, _p(ws), slab_elems, job, _stream()),
                   "conv2d_wgrad_slabs_rows")
    slab_jobs.append((job, 1, ws))
    return sums


def conv2d_wgrad_batched(items, hw, ksize, flags=0, scale=1.0, slab_jobs=None):
    """items: [(x, dy, dw, dbias | None)] of identical geometry; ACCUMULATES every dw (and dbias) in as few launches
    as possible.  slab_jobs: a list -- where the geometry has a slab form the partial tiles are written to slabs instead of
    added with fp32 atomics and the jobs that sum them into the dw are APPENDED to the list: the caller owes a sum_slabs(list)."""
    x0, dy0 = items[0][0], items[0][1]
    n, cin, cout = x0.shape[0], x0.shape[3], dy0.shape[3]
    table = (WgradItem * len(items))()
    for t, (x, dy, dw, db) in zip(table, items):
        assert x.shape == x0.shape and dy.shape == dy0.shape and dw.numel() == ksize * ksize * cin * cout
        t.x, t.dy, t.dw = _p(x, BF16, "x").value, _p(dy, BF16, "dy").value, _p(dw, F32, "dw").value
        t.dbias = _p(db, F32, "dbias").value if db is not None else None
    if slab_jobs is not None:
        ws_elems = int(lib().gank_conv2d_wgrad_batched_ws_elems(len(items), n, hw[0], hw[1], cin, cout, ksize, flags))
        if ws_elems > 0:
            ws = torch.empty(ws_elems, dtype=F32, device=x0.device)
            jobs = (SlabJob * len(items))()
            _lib.check(lib().gank_conv2d_wgrad_batched_slabs(table, len(items), n, hw[0], hw[1], cin, cout, ksize, flags, scale, _p(ws), ws_elems,
                                                             jobs, _stream()), "conv2d_wgrad_batched_slabs")
            slab_jobs.append((jobs, len(items), ws))
            return
    _lib.check(lib().gank_conv2d_wgrad_batched(table, len(items), n, hw[0], hw[1], cin, cout, ksize, flags, scale, _stream()),
               "conv2d_wgrad_batched")


def slab_job(slabs, out, n, stride, nslabs, scale=1.0):
    """one gank_slab_job as an entry of a slab-job list: out[:n] += scale * sum_s slabs.view(-1)[s * stride : s * stride + n]"""
    jobs = (SlabJob * 1)()
    jobs[0].slabs, jobs[0].out, jobs[0].n, jobs[0].stride, jobs[0].nslabs, jobs[0].scale = _p(slabs, F32, "slabs").value, _p(out, F32, "out").value, n, stride, nslabs, scale
    return (jobs, 1, slabs)


SUM_SLABS_MAX_JOBS = 12       # jobs of one launch (gank_sum_slabs_label_bwd takes no more)


def sum_slabs(slab_jobs, label=None):
    """ONE launch (per 12 jobs) for every job of the list (entries: (SlabJob array, count, the workspace kept alive)); clears the list.
    label = (sums, lists, t, w, c0, dw, g_pooled, c0g, n): label_conv3x3_bwd_pooled's launch rides on this one as extra workgroups
    (gank_sum_slabs_label_bwd; at most 12 jobs) -> its de_parts fp32 [10,V,C2]"""
    total = sum(c for _, c, _ in slab_jobs)
    if total == 0:
        assert label is None
        return None
    table, i = (SlabJob * total)(), 0
    for jobs, c, _ in slab_jobs:
        for j in range(c):
            for f, _t in SlabJob._fields_:
                setattr(table[i], f, getattr(jobs[j], f))
            i += 1
    parts = None
    if label is not None:
        sums, lists, t, w, c0, dw, g_pooled, c0g, n = label
        assert total <= SUM_SLABS_MAX_JOBS and dw.shape == w.shape and g_pooled.shape[0] == n
        v, c2 = t.shape
        hwp, pitch = g_pooled.shape[1] * g_pooled.shape[2], g_pooled.shape[3]
        parts = torch.empty((10, v, c2), dtype=F32, device=g_pooled.device)
        _lib.check(lib().gank_sum_slabs_label_bwd(table, total, _p(sums, F32, "sums"), _p(lists, I32, "lists"), _p(t, BF16, "T"), v, _p(w, F32, "w"), w.shape[2], c0,
                                                  c2, w.shape[3], n, _p(dw, F32, "dw"), _p(parts), _p(g_pooled, BF16, "g_pooled"), hwp, pitch, c0g, _stream()),
                   "sum_slabs_label_bwd")
    else:
        _lib.check(lib().gank_sum_slabs(table, total, _stream()), "sum_slabs")
    slab_jobs.clear()
    return parts


def conv2d_wgrad_narrow_pair(a, b):
    """a, b = (x, dy, dw, dbias | None, (H, W), ksize): two filter gradients of 3-channel-input layers, ACCUMULATED, one launch"""
    args = []
    for x, dy, dw, db, hw, k in (a, b):
        assert x.shape[3] == 3 and dw.numel() == k * k * 3 * dy.shape[3], (x.shape, dy.shape, dw.shape)
        args += [_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(db, F32, "dbias"), x.shape[0], hw[0], hw[1], dy.shape[3], k]
    _lib.check(lib().gank_conv2d_wgrad_narrow_pair(*args, 1.0, _stream()), "conv2d_wgrad_narrow_pair")


def conv2d_general_fprop(x, wf, bias, out_hw, cout, ksize, stride, pad, flags=0):
    """any filter size / stride 1|2 / leading pad (gank_conv2d_general_fprop); x as stored [N,Hin,Win,Cin]"""
    n, hin, win, cin = x.shape
    y = torch.empty((n, out_hw[0], out_hw[1], cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_conv2d_general_fprop(_p(x, BF16, "x"), _p(wf, BF16, "wf"), _p(bias, F32, "bias"), _p(y), n, hin, win,
                                               out_hw[0], out_hw[1], cin, cout, ksize, stride, pad, flags, _stream()), "conv2d_general_fprop")
    return y


def conv2d_general_dgrad(dy, wd, x_hw, cin, ksize, pad, relu_ref=None):
    """stride-1 input gradient at size x_hw (the gathered size: 2x the stored one for an upsampled-input conv)"""
    n, hdy, wdy, cout = dy.shape
    dx = torch.empty((n, x_hw[0], x_hw[1], cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_conv2d_general_dgrad(_p(dy, BF16, "dy"), _p(wd, BF16, "wd"), _p(relu_ref, BF16, "relu_ref"), _p(dx), n, x_hw[0], x_hw[1],
                                               hdy, wdy, cin, cout, ksize, pad, _stream()), "conv2d_general_dgrad")
    return dx


def im2col_narrow(x, out_hw, ksize, stride, pad, kpad):
    """[N,Hin,Win,Cin] -> [N,Ho,Wo,kpad] bf16, column tap*Cin + c (gank_im2col_narrow)"""
    n, hin, win, cin = x.shape
    y = torch.empty((n, out_hw[0], out_hw[1], kpad), dtype=BF16, device=x.device)
    _lib.check(lib().gank_im2col_narrow(_p(x, BF16, "x"), _p(y), n, hin, win, cin, out_hw[0], out_hw[1], ksize, stride, pad, kpad, _stream()), "im2col_narrow")
    return y


def tap_gather_up2(Z, bias, ksize, pad, cout, tanh_out=False):
    """Z [N,h,w,Zc] (column t*cout + co) -> y [N,2h,2w,cout] = (tanh)(bias + taps gathered): gank_tap_gather_up2"""
    n, h, w, zc = Z.shape
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=BF16, device=Z.device)
    _lib.check(lib().gank_tap_gather_up2(_p(Z, BF16, "Z"), _p(bias, F32, "bias"), _p(y), n, h, w, ksize, pad, cout, zc, 1 if tanh_out else 0, _stream()),
               "tap_gather_up2")
    return y


def tap_scatter_up2(g, ksize, pad, zc):
    """g [N,2h,2w,cout] -> col [N,h,w,zc]: the gradient of Z (gank_tap_scatter_up2)"""
    n, h2, w2, cout = g.shape
    col = torch.empty((n, h2 // 2, w2 // 2, zc), dtype=BF16, device=g.device)
    _lib.check(lib().gank_tap_scatter_up2(_p(g, BF16, "g"), _p(col), n, h2 // 2, w2 // 2, ksize, pad, cout, zc, _stream()), "tap_scatter_up2")
    return col


def zeros_f32(shape, device):
    """fp32 zeros by a kernel of the library (scratch gradients)"""
    t = torch.empty(shape, dtype=F32, device=device)
    _lib.check(lib().gank_zero_f32(_p(t), t.numel(), _stream()), "zero_f32")
    return t


def zero_(t):
    """t <- 0 in place by a kernel of the library (fp32, contiguous: gradient buffers)"""
    _lib.check(lib().gank_zero_f32(_p(t, F32, "t"), t.numel(), _stream()), "zero_f32")
    return t


def phase_stack4(w4):
    """[4,4,Cin,Cout] fp32 -> the stacked 3x3 filter [3,3,Cin,4*Cout] (gank_phase_stack4)"""
    cin, cout = w4.shape[2], w4.shape[3]
    w3 = torch.empty((3, 3, cin, 4 * cout), dtype=F32, device=w4.device)
    _lib.check(lib().gank_phase_stack4(_p(w4, F32, "w4"), _p(w3), cin, cout, 0, _stream()), "phase_stack4")
    return w3


def phase_stack4_bwd(g3, dw4):
    """dw4 [4,4,Cin,Cout] += adjoint of phase_stack4 applied to g3 [3,3,Cin,4*Cout]"""
    cin, cout = dw4.shape[2], dw4.shape[3]
    _lib.check(lib().gank_phase_stack4(_p(dw4, F32, "dw4"), _p(g3, F32, "g3"), cin, cout, 1, _stream()), "phase_stack4_bwd")
    return dw4


def pad_rows(src, w_out):
    """[..., w_in] -> [..., w_out] with zeros behind (fp32 or the 16-bit activation dtype)"""
    w_in = src.shape[-1]
    dst = torch.empty(tuple(src.shape[:-1]) + (w_out,), dtype=src.dtype, device=src.device)
    _lib.check(lib().gank_pad_rows(_p(src), _p(dst), src.numel() // w_in, w_in, w_out, src.element_size(), 0, _stream()), "pad_rows")
    return dst


def pad_rows_bwd(g, dst):
    """dst [..., w_in] (+)= g [..., w_out][..., :w_in]  (fp32 accumulates, 16-bit overwrites)"""
    w_in, w_out = dst.shape[-1], g.shape[-1]
    _lib.check(lib().gank_pad_rows(_p(g), _p(dst), dst.numel() // w_in, w_in, w_out, g.element_size(), 1, _stream()), "pad_rows_bwd")
    return dst


def tile_rows(b, reps):
    out = torch.empty(reps * b.numel(), dtype=F32, device=b.device)
    _lib.check(lib().gank_tile_rows(_p(b, F32, "b"), _p(out), reps, b.numel(), 0, _stream()), "tile_rows")
    return out


def tile_rows_bwd(g, db, reps):
    _lib.check(lib().gank_tile_rows(_p(g, F32, "g"), _p(db, F32, "db"), reps, db.numel(), 1, _stream()), "tile_rows_bwd")
    return db


def fewout_pack(w, zc):
    """[k,k,Cin,Cout<=4] -> [Cin, zc] with column t*Cout + co"""
    k, cin, cout = w.shape[0], w.shape[2], w.shape[3]
    wz = torch.empty((cin, zc), dtype=F32, device=w.device)
    _lib.check(lib().gank_fewout_pack(_p(w, F32, "w"), _p(wz), k, cin, cout, zc, 0, _stream()), "fewout_pack")
    return wz


def fewout_pack_bwd(gz, dw):
    k, cin, cout = dw.shape[0], dw.shape[2], dw.shape[3]
    _lib.check(lib().gank_fewout_pack(_p(gz, F32, "gz"), _p(dw, F32, "dw"), k, cin, cout, gz.shape[1], 1, _stream()), "fewout_pack_bwd")
    return dw


def depth_to_space2(x):
    """[N,h,w,4C] -> [N,2h,2w,C], channel order (a, b, c)"""
    n, h, w, c4 = x.shape
    y = torch.empty((n, 2 * h, 2 * w, c4 // 4), dtype=BF16, device=x.device)
    _lib.check(lib().gank_depth_to_space2(_p(x, BF16, "x"), _p(y), n, h, w, c4 // 4, _stream()), "depth_to_space2")
    return y


def space_to_depth2(y):
    """[N,2h,2w,C] -> [N,h,w,4C]: the adjoint (and inverse) of depth_to_space2"""
    n, h2, w2, c = y.shape
    x = torch.empty((n, h2 // 2, w2 // 2, 4 * c), dtype=BF16, device=y.device)
    _lib.check(lib().gank_space_to_depth2(_p(y, BF16, "y"), _p(x), n, h2 // 2, w2 // 2, c, _stream()), "space_to_depth2")
    return x


IM2COL_NARROW_WGRAD = True    # filter gradients of layers with k*k*Cin <= 128 and Cin < 32 (Pix2Pix's 4x4 stride-2 input layers) through im2col + the 1x1 MFMA kernels


def conv2d_general_wgrad(x, dy, dw, ksize, stride, pad, flags=0, dbias=None):
    """ACCUMULATES into dw fp32 [k,k,Cin,Cout] (and dbias)"""
    n, hx, wx, cin = x.shape
    _, hdy, wdy, cout = dy.shape
    assert dw.numel() == ksize * ksize * cin * cout
    ktot = ksize * ksize * cin
    if IM2COL_NARROW_WGRAD and flags == 0 and cin < 32 and ksize > 1 and ktot <= 128 and cout % 64 == 0 and n * hdy * wdy >= 16384:
        kpad = 64 if ktot <= 64 else 128
        xcol = im2col_narrow(x, (hdy, wdy), ksize, stride, pad, kpad)
        tmp = zeros_f32((1, 1, kpad, cout), x.device)
        conv2d_wgrad(xcol, dy, tmp, (hdy, wdy), 1, 0, 1.0, dbias=dbias)
        dwv = dw.view(ktot, cout)
        weighted_sum_f32([dwv, tmp.view(kpad, cout)[:ktot]], [1.0, 1.0], out=dwv)     # the rows of the real taps; dw ACCUMULATES
        return dw
    _lib.check(lib().gank_conv2d_general_wgrad(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"), n, hx, wx, hdy, wdy,
                                               cin, cout, ksize, stride, pad, flags, _stream()), "conv2d_general_wgrad")
    return dw


def upconv3x3_prep(w):
    """w fp32 [3,3,Cin,Cout] -> (wph, wd4) for the phase-decomposed NN-upsample+3x3 conv; cached on the tensor
    as `w._prep_up` and rewritten IN PLACE on later calls (captured graphs keep reading the same buffers)."""
    _, _, cin, cout = w.shape
    old = getattr(w, "_prep_up", None)
    if old is not None:
        wph, wd4 = old
    else:
        wph = torch.empty((4, _roundup(cout, 32), 4 * cin), dtype=BF16, device=w.device)
        wd4 = torch.empty((_roundup(cin, 32), _roundup(16 * cout, 64)), dtype=BF16, device=w.device)
    _lib.check(lib().gank_upconv3x3_prep_weights(_p(w.detach(), F32, "w"), _p(wph), _p(wd4), cin, cout, _stream()), "upconv3x3_prep")
    w._prep_up = (wph, wd4)
    return wph, wd4


def upconv3x3_fprop(x, wph, bias, cout, flags=0, residual=None, stats_groups=0):
    n, hl, wl, cin = x.shape
    y = torch.empty((n, 2 * hl, 2 * wl, cout), dtype=BF16, device=x.device)
    if stats_groups:
        sums, pre = stats_arena.take((stats_groups, _lib.STAT_SLOTS, 2, cout), x.device)
        produced = C.c_int(0)
        _lib.check(lib().gank_upconv3x3_fprop_stats(_p(x, BF16, "x"), _p(wph, BF16), _p(bias, F32, "bias"), _p(residual, BF16, "residual"),
                                                    _p(y), n, hl, wl, cin, cout, flags | (STATS_PREZEROED if pre else 0), _p(sums), stats_groups, C.byref(produced), _stream()),
                   "upconv3x3_fprop_stats")
        return y, (ConvStats(sums, bias, stats_groups) if produced.value else None)
    _lib.check(lib().gank_upconv3x3_fprop(_p(x, BF16, "x"), _p(wph, BF16), _p(bias, F32, "bias"), _p(residual, BF16, "residual"),
                                          _p(y), n, hl, wl, cin, cout, flags, _stream()), "upconv3x3_fprop")
    return y


def upconv3x3_dgrad(dy, wd4, cin, relu_ref=None):
    n, h2, w2, cout = dy.shape
    dx = torch.empty((n, h2 // 2, w2 // 2, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_upconv3x3_dgrad(_p(dy, BF16, "dy"), _p(wd4, BF16), _p(relu_ref, BF16, "relu_ref"), _p(dx),
                                          n, h2 // 2, w2 // 2, cin, cout, _stream()), "upconv3x3_dgrad")
    return dx


def convpool3x3_prep(w):
    """w fp32 [3,3,Cin,Cout] -> (wp4, wphd): operands of ConvMeanPool 3x3 run as one 4x4 stride-2 conv; cached on the
    tensor as `w._prep_pool` and rewritten IN PLACE on later calls."""
    _, _, cin, cout = w.shape
    old = getattr(w, "_prep_pool", None)
    if old is not None:
        wp4, wphd = old
    else:
        wp4 = torch.empty((_roundup(cout, 32), _roundup(16 * cin, 64)), dtype=BF16, device=w.device)
        wphd = torch.empty((4, _roundup(cin, 32), 4 * cout), dtype=BF16, device=w.device)
    _lib.check(lib().gank_convpool3x3_prep_weights(_p(w.detach(), F32, "w"), _p(wp4), _p(wphd), cin, cout, _stream()), "convpool3x3_prep")
    w._prep_pool = (wp4, wphd)
    return wp4, wphd


def convpool3x3_fprop(x, wp4, bias, cout, flags=0, residual=None):
    n, h, w, cin = x.shape
    y = torch.empty((n, h // 2, w // 2, cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_convpool3x3_fprop(_p(x, BF16, "x"), _p(wp4, BF16), _p(bias, F32, "bias"), _p(residual, BF16, "residual"),
                                            _p(y), n, h // 2, w // 2, cin, cout, flags, _stream()), "convpool3x3_fprop")
    return y


def convpool3x3_dgrad(dy, wphd, cin, relu_ref=None):
    n, hp, wp, cout = dy.shape
    dx = torch.empty((n, 2 * hp, 2 * wp, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_convpool3x3_dgrad(_p(dy, BF16, "dy"), _p(wphd, BF16), _p(relu_ref, BF16, "relu_ref"), _p(dx),
                                            n, hp, wp, cin, cout, _stream()), "convpool3x3_dgrad")
    return dx


def cpool_res_ok(n, hp, wp, cin, cout):
    """shapes the resident ConvMeanPool kernels cover (fprop and dgrad)"""
    return cout == 128 and cin % 128 == 0 and hp % 8 == 0 and (wp % 16 == 0 or wp == 8) and n * 4 * hp * wp * cin < (1 << 30)


def res8_conv3x3_ok(n, hw, cin, cout):
    """geometry of gank_res8_conv3x3: 8x8 images (after the upsample, if any), Cin 128 | 256, Cout % 128 == 0"""
    return tuple(hw) == (8, 8) and cin in (128, 256) and cout % 128 == 0 and n * 64 * max(cin, cout) < (1 << 30)


def img16_conv3x3_ok(n, hw, cin, cout):
    """geometry of gank_img16_conv3x3: 16x16 images, Cin % 64 == 0, Cout % 128 == 0"""
    return tuple(hw) == (16, 16) and cin % 64 == 0 and cout % 128 == 0 and n * 256 * max(cin, cout) < (1 << 30)


def img16_conv3x3(x, rf, bias, cout, flags=0, relu_ref=None, residual=None, stats_groups=0):
    """3x3 SAME conv on LDS-resident 16x16 images (rf: prep kind 4 operand, rows = output channels); flags: IN_RELU, RES_UPSAMPLE2X
    (residual is [N,8,8,Cout]).  stats_groups > 0 -> (y, ConvStats)"""
    n, cin = x.shape[0], x.shape[3]
    assert tuple(x.shape[1:3]) == (16, 16), x.shape
    y = torch.empty((n, 16, 16, cout), dtype=BF16, device=x.device)
    if stats_groups:
        sums, pre = stats_arena.take((stats_groups, _lib.STAT_SLOTS, 2, cout), x.device)
        _lib.check(lib().gank_img16_conv3x3_stats(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(relu_ref, BF16, "relu_ref"),
                                                  _p(residual, BF16, "residual"), _p(y), n, cin, cout, int(flags) | (STATS_PREZEROED if pre else 0),
                                                  _p(sums), stats_groups, _stream()), "img16_conv3x3_stats")
        return y, ConvStats(sums, bias, stats_groups)
    _lib.check(lib().gank_img16_conv3x3(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(relu_ref, BF16, "relu_ref"),
                                        _p(residual, BF16, "residual"), _p(y), n, cin, cout, int(flags), _stream()), "img16_conv3x3")
    return y


def cbn_relu_img16_conv3x3(x, labels, gamma, beta, stats, rf, bias, cout, flags=0, residual=None, stats_groups=0):
    """conv3x3_SAME(relu(cond_batchnorm(x))) + bias (+ residual) on 16x16 images, the normalisation inside the image-resident
    kernel's operand staging (no grad; stats [groups,2,Cin] from cbn_stats).  flags: RES_UPSAMPLE2X.  stats_groups > 0 -> (y, ConvStats)"""
    n, cin = x.shape[0], x.shape[3]
    assert tuple(x.shape[1:3]) == (16, 16), x.shape
    groups = stats.shape[0]
    y = torch.empty((n, 16, 16, cout), dtype=BF16, device=x.device)
    sums, pre = (stats_arena.take((stats_groups, _lib.STAT_SLOTS, 2, cout), x.device) if stats_groups else (None, False))
    _lib.check(lib().gank_cbn_relu_img16_conv3x3(_p(x, BF16, "x"), _p(labels, I32, "labels"), _p(gamma, F32, "gamma"), _p(beta, F32, "beta"),
                                                 _p(stats, F32, "stats"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(residual, BF16, "residual"),
                                                 _p(y), n, cin, cout, groups, gamma.shape[0], int(flags) | (STATS_PREZEROED if pre else 0),
                                                 _p(sums), stats_groups, _stream()), "cbn_relu_img16_conv3x3")
    if stats_groups:
        return y, ConvStats(sums, bias, stats_groups)
    return y


def res8_conv3x3(x, rf, bias, cout, flags=0, residual=None, stats_groups=0):
    """3x3 SAME conv on LDS-resident 8x8 images (rf: prep kind 4 operand, rows = output channels).  flags: IN_UPSAMPLE2X
    (x is [N,4,4,Cin]), RES_UPSAMPLE2X (residual is [N,4,4,Cout]), OUT_POOLSUM2X (result as 2x2 sums [N,4,4,Cout]).
    stats_groups > 0 -> (y, ConvStats)"""
    n, cin = x.shape[0], x.shape[3]
    hw = 4 if flags & OUT_POOLSUM2X else 8
    y = torch.empty((n, hw, hw, cout), dtype=BF16, device=x.device)
    if stats_groups:
        sums, pre = stats_arena.take((stats_groups, _lib.STAT_SLOTS, 2, cout), x.device)
        _lib.check(lib().gank_res8_conv3x3(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(residual, BF16, "residual"), _p(y),
                                           n, cin, cout, flags | (STATS_PREZEROED if pre else 0), _p(sums), stats_groups, _stream()), "res8_conv3x3")
        return y, ConvStats(sums, bias, stats_groups)
    _lib.check(lib().gank_res8_conv3x3(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(residual, BF16, "residual"), _p(y),
                                       n, cin, cout, flags, None, 0, _stream()), "res8_conv3x3")
    return y


def cpool_res_fprop(x, rf, bias, cout, flags=0, residual=None):
    n, h, w, cin = x.shape
    y = torch.empty((n, h // 2, w // 2, cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_cpool_res_fprop(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias, F32, "bias"), _p(residual, BF16, "residual"),
                                          _p(y), n, h // 2, w // 2, cin, cout, flags, _stream()), "cpool_res_fprop")
    return y


def cpool_res_dgrad(dy, rd, cin, relu_ref=None):
    n, hp, wp, cout = dy.shape
    dx = torch.empty((n, 2 * hp, 2 * wp, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_cpool_res_dgrad(_p(dy, BF16, "dy"), _p(rd, BF16, "rd"), _p(relu_ref, BF16, "relu_ref"), _p(dx),
                                          n, hp, wp, cin, cout, _stream()), "cpool_res_dgrad")
    return dx


def cpool_res_dgrad_image_wgrad_ok(dy, cin):
    """geometry of gank_cpool_res_dgrad_image_wgrad: 16-wide pooled grid (32-pixel image rows), 128 dy channels"""
    n, hp, wp, cout = dy.shape
    return wp == 16 and hp % 8 == 0 and cout == 128 and cin % 128 == 0 and n * 4 * hp * wp * cin < (1 << 30)


def cpool_res_dgrad_image_wgrad(dy, rd, relu_ref, x_image, dw1, db1=None, x_pooled=None, dws=None, dbs=None, slab_jobs=None):
    """ConvMeanPool input gradient whose only consumer is the filter gradient of the 3-channel-input 3x3 conv in front: nothing is
    stored; dw1 fp32 [3,3,3,Cin] / db1 [Cin] (and, with x_pooled, the 1x1 shortcut's dws [1,1,3,Cout] / dbs) are ACCUMULATED --
    by fp32 atomics, or (slab_jobs: a list, Cin == 128) each workgroup's tile goes to a slab of its own and the jobs that add the
    slabs into the targets are appended to the list: the caller owes a sum_slabs(list)."""
    n, hp, wp, cout = dy.shape
    cin = relu_ref.shape[3]
    assert tuple(relu_ref.shape) == (n, 2 * hp, 2 * wp, cin) and tuple(x_image.shape) == (n, 2 * hp, 2 * wp, 3), (relu_ref.shape, x_image.shape)
    assert dw1.numel() == 27 * cin and (db1 is None or db1.numel() == cin)
    assert x_pooled is None or (tuple(x_pooled.shape) == (n, hp, wp, 3) and dws is not None and dws.numel() == 3 * cout)
    slabs = None
    if slab_jobs is not None and cin == 128:
        nsl = n * (hp // 8)
        slabs = torch.empty(nsl * 4096, dtype=F32, device=dy.device)
    _lib.check(lib().gank_cpool_res_dgrad_image_wgrad(_p(dy, BF16, "dy"), _p(rd, BF16, "rd"), _p(relu_ref, BF16, "relu_ref"),
                                                      _p(x_image, BF16, "x_image"), _p(dw1, F32, "dw1"), _p(db1, F32, "db1"),
                                                      _p(x_pooled, BF16, "x_pooled"), _p(dws, F32, "dws"), _p(dbs, F32, "dbs"),
                                                      n, hp, wp, cin, cout, _p(slabs), _stream()), "cpool_res_dgrad_image_wgrad")
    if slabs is not None:
        for row0, rows, tgt in ((0, 27, dw1), (27, 1, db1), (28, 3, dws if x_pooled is not None else None), (31, 1, dbs if x_pooled is not None else None)):
            if tgt is not None:
                slab_jobs.append(slab_job(slabs[row0 * 128:], tgt, rows * 128, 4096, nsl))


def convpool3x3_wgrad(x, dy, dw, flags=0, dbias=None, slab_jobs=None):
    """ACCUMULATES the ConvMeanPool 3x3 filter gradient into dw fp32 [3,3,Cin,Cout] (and dbias).  slab_jobs: a list -- the
    sum-and-fold of the filter-row kernel's slabs is left to the caller's sum_slabs(list) (job appended; dbias as before)."""
    n, hp, wp, cout = dy.shape
    cin = x.shape[3]
    assert x.shape[1] == 2 * hp and x.shape[2] == 2 * wp and dw.numel() == 9 * cin * cout, (x.shape, dy.shape, dw.shape)
    ws_elems = lib().gank_convpool3x3_wgrad_ws_elems(n, hp, wp, cin, cout)
    ws16 = torch.empty(ws_elems, dtype=F32, device=x.device)
    if slab_jobs is not None:
        job = (SlabJob * 1)()
        _lib.check(lib().gank_convpool3x3_wgrad_job(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"),
                                                    _p(ws16), ws_elems, n, hp, wp, cin, cout, flags, job, _stream()), "convpool3x3_wgrad_job")
        if job[0].nslabs > 0:
            slab_jobs.append((job, 1, ws16))
        return
    _lib.check(lib().gank_convpool3x3_wgrad(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(dbias, F32, "dbias"),
                                            _p(ws16), ws_elems, n, hp, wp, cin, cout, flags, _stream()), "convpool3x3_wgrad")
    return dw


def upconv3x3_wgrad_ws(xl, cout):
    """workspace floats of upconv3x3_wgrad for this input; 0: the shape is not served by the phase form"""
    n, h, w, cin = xl.shape
    return int(lib().gank_upconv3x3_wgrad_ws_elems(n, h, w, cin, cout))


def upconv3x3_wgrad(xl, dy, dw):
    """ACCUMULATES the UpsampleConv 3x3 filter gradient into dw fp32 [3,3,Cin,Cout]: xl = the conv's input before the upsample
    [N,H,W,Cin], dy [N,2H,2W,Cout] (phase form on the ConvMeanPool rows kernel: gank_upconv3x3_wgrad).  No bias gradient."""
    n, h, w, cin = xl.shape
    cout = dy.shape[3]
    assert dy.shape[1] == 2 * h and dy.shape[2] == 2 * w and dw.numel() == 9 * cin * cout, (xl.shape, dy.shape, dw.shape)
    ws_elems = upconv3x3_wgrad_ws(xl, cout)
    if ws_elems <= 0:
        raise RuntimeError(f"gank: upconv3x3_wgrad does not serve {tuple(xl.shape)} -> {cout} channels")
    ws16 = torch.empty(ws_elems, dtype=F32, device=xl.device)
    _lib.check(lib().gank_upconv3x3_wgrad(_p(xl, BF16, "x_low"), _p(dy, BF16, "dy"), _p(dw, F32, "dw"), _p(ws16), ws_elems, n, h, w, cin, cout, _stream()),
               "upconv3x3_wgrad")
    return dw


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def res8_chain_fwd(x, w_rfrag, biases, keep=True, pool=False, head=None):
    """Fused identity-shortcut residual blocks on 8x8 images (gank_res8_chain_fwd).  x bf16 [N,8,8,128]; w_rfrag: the
    2*nblocks kind-4 `rf` operands (conv_1, conv_2 per block); biases: fp32 [128] or None each.
    -> (out, h1s, ys): out = pooled [N,128] when pool else the last block's output; h1s / ys = per-block tensors kept for
    the backward pass (empty lists when keep=False and they are not the result).
    head = (w fp32 [128], b fp32 [1] | None) with pool: the critic's last dense layer inside the launch (gank_res8_chain_fwd_head);
    -> (pooled, h1s, ys, logits bf16 [N])"""
    n, hh, ww, c = x.shape
    assert hh == 8 and ww == 8 and len(w_rfrag) in (2, 4) and len(biases) == len(w_rfrag), (x.shape, len(w_rfrag))
    nb = len(w_rfrag) // 2
    for w in w_rfrag:
        _p(w, BF16, "w_rfrag")
    for b in biases:
        _p(b, F32, "bias")
    h1s = [torch.empty_like(x) for _ in range(nb)] if keep else [None] * nb
    ys = [torch.empty_like(x) if (keep or (b == nb - 1 and not pool)) else None for b in range(nb)]
    pooled = torch.empty((n, c), dtype=BF16, device=x.device) if pool else None
    if head is not None:
        assert pool and head[0].numel() == c and (head[1] is None or head[1].numel() == 1)
        logits = torch.empty(n, dtype=BF16, device=x.device)
        _lib.check(lib().gank_res8_chain_fwd_head(_p(x, BF16, "x"), _ptr_array(w_rfrag), _ptr_array(biases), _ptr_array(h1s), _ptr_array(ys),
                                                  _p(pooled), _p(head[0], F32, "head_w"), _p(head[1], F32, "head_b"), _p(logits), n, c, nb,
                                                  _stream()), "res8_chain_fwd_head")
        return pooled, h1s, ys, logits
    _lib.check(lib().gank_res8_chain_fwd(_p(x, BF16, "x"), _ptr_array(w_rfrag), _ptr_array(biases), _ptr_array(h1s), _ptr_array(ys),
                                         _p(pooled), n, c, nb, _stream()), "res8_chain_fwd")
    return (pooled if pool else ys[-1]), h1s, ys


def res8_chain_bwd(dy, dpool, ylast, wd_rfrag, h1s, xins, keep=True, head=None):
    """Backward chain (gank_res8_chain_bwd); lists are in FORWARD order (reversed here).  dy [N,8,8,128] or None with
    dpool [N,128] + ylast.  -> (dx, g1s, dys): dx = gradient of the chain input; g1s[b] = gradient of block b's conv_1
    output, dys[b] = gradient of block b's output (forward order; None entries when keep=False).
    head = dict(logits, w, pooled, loss, w_grad | None, b_grad | None, n_real, mode, loss_scale) instead of dy / dpool: the
    hinge loss on the fused head's logits differentiated inside the launch (gank_res8_chain_bwd_head)"""
    nb = len(h1s)
    assert len(wd_rfrag) == 2 * nb and len(xins) == nb and (dy is not None or ((dpool is not None or head is not None) and ylast is not None))
    ref = xins[0]
    n, c = ref.shape[0], ref.shape[3]
    for w in wd_rfrag:
        _p(w, BF16, "wd_rfrag")
    for t in list(h1s) + list(xins):
        _p(t, BF16, "mask tensor")
    g1s = [torch.empty_like(ref) if keep else None for _ in range(nb)]
    dxs = [torch.empty_like(ref) if (keep or b == 0) else None for b in range(nb)]       # dxs[b] = gradient of block b's input
    dy_out = torch.empty_like(ref) if (dy is None and keep) else None
    wd_rev, h1_rev, x_rev, g1_rev, dx_rev = [], [], [], [], []
    for b in reversed(range(nb)):
        wd_rev += [wd_rfrag[2 * b + 1], wd_rfrag[2 * b]]       # conv_2's operand first
        h1_rev.append(h1s[b]); x_rev.append(xins[b]); g1_rev.append(g1s[b]); dx_rev.append(dxs[b])
    if head is not None:
        assert dy is None and dpool is None
        hd = Res8Head()
        hd.logits, hd.head_w, hd.pooled = _p(head["logits"], BF16, "logits").value, _p(head["w"], F32, "head_w").value, _p(head["pooled"], BF16, "pooled").value
        hd.loss = _p(head["loss"], F32, "loss").value
        hd.w_grad = _p(head.get("w_grad"), F32, "w_grad").value if head.get("w_grad") is not None else None
        hd.b_grad = _p(head.get("b_grad"), F32, "b_grad").value if head.get("b_grad") is not None else None
        hd.n_real, hd.mode, hd.loss_scale = int(head["n_real"]), int(head["mode"]), float(head["loss_scale"])
        assert head["logits"].numel() == n and head["w"].numel() == c and tuple(head["pooled"].shape) == (n, c)
        _lib.check(lib().gank_res8_chain_bwd_head(C.byref(hd), _p(ylast, BF16, "ylast"), _p(dy_out), _ptr_array(wd_rev), _ptr_array(h1_rev),
                                                  _ptr_array(x_rev), _ptr_array(g1_rev), _ptr_array(dx_rev), n, c, nb, _stream()), "res8_chain_bwd_head")
    else:
        _lib.check(lib().gank_res8_chain_bwd(_p(dy, BF16, "dy"), _p(dpool, BF16, "dpool"), _p(ylast, BF16, "ylast"), _p(dy_out),
                                             _ptr_array(wd_rev), _ptr_array(h1_rev), _ptr_array(x_rev), _ptr_array(g1_rev), _ptr_array(dx_rev),
                                             n, c, nb, _stream()), "res8_chain_bwd")
    dys = [dxs[b + 1] if b + 1 < nb else (dy if dy is not None else dy_out) for b in range(nb)]
    return dxs[0], g1s, dys


def deconv2d_prep_phases(f):
    """f fp32 [k,k,Cout,Cin], k in (3, 4) -> wph bf16 [4][roundup(Cout,32)][4*Cin] for upconv3x3_fprop"""
    k, _, cout, cin = f.shape
    wph = torch.empty((4, _roundup(cout, 32), 4 * cin), dtype=BF16, device=f.device)
    _lib.check(lib().gank_deconv2d_prep_phases(_p(f, F32, "f"), _p(wph), k, cin, cout, _stream()), "deconv2d_prep_phases")
    return wph


def deconv2d_fprop(x, wz, bias, cout, ksize):
    n, h, w, cin = x.shape
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_deconv2d_fprop(_p(x, BF16, "x"), _p(wz, BF16), _p(bias, F32), _p(y), n, h, w, cin, cout, ksize,
                                         _stream()), "deconv2d_fprop")
    return y


def deconv2d_dgrad(dy, wfz, cin, ksize):
    n, h2, w2, cout = dy.shape
    dx = torch.empty((n, h2 // 2, w2 // 2, cin), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_deconv2d_dgrad(_p(dy, BF16, "dy"), _p(wfz, BF16), _p(dx), n, h2 // 2, w2 // 2, cin, cout, ksize,
                                         _stream()), "deconv2d_dgrad")
    return dx


def deconv2d_wgrad(x, dy, df, ksize):
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    _lib.check(lib().gank_deconv2d_wgrad(_p(x, BF16, "x"), _p(dy, BF16, "dy"), _p(df, F32), n, h, w, cin, cout, ksize,
                                         _stream()), "deconv2d_wgrad")
    return df


def linear_fwd(x, w, bias=None, out_f32=False):
    """y = x w + bias on the fp32 master weight (small layers): x bf16 [M,K], w fp32 [K,C] -> y bf16 [M,C] (out_f32: fp32)"""
    m, k = x.shape
    c = w.shape[1]
    assert w.shape[0] == k, (x.shape, w.shape)
    y = torch.empty((m, c), dtype=F32 if out_f32 else BF16, device=x.device)
    fn = lib().gank_linear_fwd_f32out if out_f32 else lib().gank_linear_fwd
    _lib.check(fn(_p(x, BF16, "x"), _p(w, F32, "w"), _p(bias, F32, "bias"), _p(y), m, k, c, _stream()), "linear_fwd")
    return y


def linear_bwd(dy, x, w, want_dx=True, dw=None, dbias=None):
    """dx = dy w^T (returned when want_dx); dw += x^T dy; dbias += colsum(dy)"""
    m, c = dy.shape
    k = w.shape[0] if w is not None else (x.shape[1] if x is not None else 1)
    dx = torch.empty((m, k), dtype=BF16, device=dy.device) if want_dx else None
    _lib.check(lib().gank_linear_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(w, F32, "w"), _p(dx), _p(dw, F32, "dw"),
                                     _p(dbias, F32, "dbias"), m, k, c, _stream()), "linear_bwd")
    return dx


def colsum(x2d, out, scale=1.0):
    """out[c] += scale * sum_r x[r,c];  x bf16 [rows,C] (any leading dims flattened), out fp32 [C]"""
    c = x2d.shape[-1]
    rows = x2d.numel() // c
    _lib.check(lib().gank_colsum_bf16(_p(x2d, BF16, "x"), _p(out, F32, "out"), rows, c, scale, _stream()), "colsum")
    return out


# ------------------------------------------------------------------ spectral norm
class SnState:
    """PERSISTENT spectral-norm workspaces of one network (a, b, v, ga, scal, bpart, the u snapshot and a staging buffer for
    u'), owned by a trainer, so that the end of one update can run the power iteration of the NEXT forward pass
    (gank_sn_adam_fwd_a: spectral-norm backward + Adam + forward A in one launch).  `valid`: the workspaces hold the power
    iteration of the CURRENT (W, u) and `u_next` its u'; a forward pass then runs its second launch only, and when it is one
    that assigns u (update_collection=None) it adopts u' and clears the flag.  Only the fused tail and `refresh()` set it;
    whoever changes the weights or u behind the trainer's back (load_state_dict, tests writing into the variables) clears it
    (`SNGANTrainer.sn_state_changed`).  `u_flat`: the concatenated u vectors (ParamStore.flatten_state), `us` its views."""

    def __init__(self, weights, us, u_flat):
        self.weights, self.us, self.u_flat = list(weights), [u.detach() for u in us], u_flat
        dev = self.weights[0].device
        self.n = len(self.weights)
        assert self.n <= 16
        self.KC = [(w.numel() // w.shape[-1], w.shape[-1]) for w in self.weights]
        kt, ct = sum(k for k, _ in self.KC), sum(c for _, c in self.KC)
        assert ct == u_flat.numel() and all(u.data_ptr() == u_flat.data_ptr() + 4 * o for u, o in zip(self.us, self._coffs()))
        self.v, self.a, self.ga = (torch.empty(kt, dtype=F32, device=dev) for _ in range(3))
        self.b, self.u_snap, self.u_next = (torch.empty(ct, dtype=F32, device=dev) for _ in range(3))
        self.scal = torch.zeros(self.n * 8, dtype=F32, device=dev)
        self.ws = [int(lib().gank_sn_ws_floats(k, c)) for k, c in self.KC]
        self.bpart = torch.empty(sum(self.ws), dtype=F32, device=dev)
        self.valid = False
        self._key = tuple(w.data_ptr() for w in self.weights) + tuple(u.data_ptr() for u in self.us)

    def _coffs(self):
        out, o = [], 0
        for _, c in self.KC:
            out.append(o)
            o += c
        return out

    def matches(self, weights, us):
        return len(weights) == self.n and tuple(w.data_ptr() for w in weights) + tuple(u.data_ptr() for u in us) == self._key

    def refresh(self):
        """forward A alone on the current (W, u): u' -> u_next, nothing else moves"""
        SnBatch(self.weights, self.us, state=self).forward_a_only()
        self.valid = True


class SnBatch:
    """Workspaces + descriptor table for one batched spectral-norm call over several weights."""

    def __init__(self, weights, us, snapshot=False, inplace=False, state=None):
        """snapshot: the kernels keep their own copy of u for the backward pass; inplace: u_final is written straight
        over `us` (needs snapshot when a backward pass follows): u.assign(u_final) without clone/copy launches.
        state: an SnState for exactly these weights and u vectors -- its persistent workspaces are used instead of fresh ones."""
        self.weights, self.us = list(weights), list(us)
        dev = self.weights[0].device
        self.n = len(self.weights)
        self.KC = [(w.numel() // w.shape[-1], w.shape[-1]) for w in self.weights]
        tot = lambda f: sum(f(k, c) for k, c in self.KC)  # noqa: E731
        self.W_bar = [torch.empty_like(w) for w in self.weights]
        self.state = state if (state is not None and state.matches(self.weights, self.us)) else None
        ws = [int(lib().gank_sn_ws_floats(k, c)) for k, c in self.KC]
        if self.state is not None:
            st = self.state
            # u' of a pass that does not assign it goes to the staging buffer (the workspaces then stay consistent with (W, u))
            self.u_out, self.v, self.a, self.b, self.scal, self.bpart, self.ga = st.u_next, st.v, st.a, st.b, st.scal, st.bpart, st.ga
            self.u_snap = st.u_snap if snapshot else None
        else:
            self.u_out = torch.empty(tot(lambda k, c: c), dtype=F32, device=dev)
            self.v = torch.empty(tot(lambda k, c: k), dtype=F32, device=dev)
            self.a = torch.empty_like(self.v)
            self.b = torch.empty_like(self.u_out)
            self.scal = torch.empty(self.n * 8, dtype=F32, device=dev)     # every entry is plainly written before it is read
            # partial column sums / |a|^2 / <G,W> per row chunk (the library says how many floats: 16-byte aligned regions)
            self.bpart = torch.empty(sum(ws), dtype=F32, device=dev)
            self.ga = torch.empty_like(self.v)
            self.u_snap = torch.empty_like(self.u_out) if snapshot else None
        self.inplace = inplace
        self.prep = None          # (kinds, want_d): MFMA operand copies of the normalised weights from the same launch pair
        self.label = None         # (table fp32 [V,D], weight index, bias | None): per-label rows of a small dense layer
        self.label_out = None
        self.table = (SnDesc * self.n)()
        ko = co = bo = 0
        for i, (w, u, (k, c)) in enumerate(zip(self.weights, self.us, self.KC)):
            d = self.table[i]
            d.W, d.u_in = _p(w, F32, "W").value, _p(u, F32, "u").value
            assert u.numel() == c, (u.shape, c)
            d.u_out = d.u_in if inplace else self.u_out.data_ptr() + 4 * co
            d.u_snap = self.u_snap.data_ptr() + 4 * co if snapshot else None
            d.v = self.v.data_ptr() + 4 * ko
            d.W_bar = self.W_bar[i].data_ptr()
            d.scal = self.scal.data_ptr() + 32 * i
            d.a = self.a.data_ptr() + 4 * ko
            d.b = self.b.data_ptr() + 4 * co
            d.bpart = self.bpart.data_ptr() + 4 * bo
            d.rowdot = None
            d.ga = self.ga.data_ptr() + 4 * ko
            d.K, d.C = k, c
            ko, co, bo = ko + k, co + c, bo + ws[i]
        self._co = co

    def forward_a_only(self):
        _lib.check(lib().gank_sn_power_iter_fwd_a(self.table, self.n, _stream()), "sn_power_iter_fwd_a")

    def forward(self):
        st = self.state
        b_only = st is not None and st.valid and self.n <= 16        # the power iteration of this (W, u) has been run already
        if st is not None:
            st.valid = b_only and not self.inplace                   # after this pass: u moved on (assigning pass), or nothing changed
        if self.prep is None and self.label is None and not b_only:
            feed = take_deferred_critic_feed()
            if feed is not None:
                critic_feed(*feed)
            _lib.check(lib().gank_sn_power_iter_fwd(self.table, self.n, _stream()), "sn_power_iter_fwd")
            return self.W_bar
        # W / sigma, its bf16 MFMA operand copies and the label table in one launch pair: the operand entries read the
        # MASTER weights (the kernel divides by sigma); the copies belong to the normalised tensors
        kinds, want_d = self.prep if self.prep is not None else ([None] * self.n, True)
        ptable, todo, outs = _prep_plan(self.W_bar, want_d, kinds, sources=self.weights)
        pw = (C.c_int * max(len(todo), 1))(*todo)
        ldesc = None
        if self.label is not None:
            tab, wi, bias = self.label
            v, dd = tab.shape
            assert self.KC[wi][0] == dd, (self.KC[wi], tab.shape)
            self.label_out = torch.empty((v, self.KC[wi][1]), dtype=BF16, device=tab.device)
            ldesc = LabelDenseDesc(_p(tab.detach(), F32, "table").value, _p(bias.detach(), F32, "bias").value if bias is not None else None,
                                   self.label_out.data_ptr(), v, dd, wi)
        feed = take_deferred_critic_feed()
        if b_only:
            # second launch only; an assigning pass adopts the staged u' (u_snap <- u, u <- u_next: flat copies); a deferred
            # critic feed (defer_critic_feed) rides on the launch as a block range of its own
            adopt = self.inplace
            args = (self.table, self.n, ptable, pw, len(todo), C.byref(ldesc) if ldesc is not None else None,
                    _p(st.u_flat, F32, "u_flat") if adopt else None,
                    _p(self.u_snap, F32, "u_snap") if (adopt and self.u_snap is not None) else None,
                    _p(st.u_next, F32, "u_next") if adopt else None, st.u_flat.numel() if adopt else 0)
            if feed is not None:
                _lib.check(lib().gank_sn_power_iter_fwd_b_prep_feed(*args, C.byref(_critic_feed_desc(*feed)), _stream()), "sn_power_iter_fwd_b_prep_feed")
            else:
                _lib.check(lib().gank_sn_power_iter_fwd_b_prep(*args, _stream()), "sn_power_iter_fwd_b_prep")
        else:
            if feed is not None:
                critic_feed(*feed)        # no launch to ride on: the feed's own
            _lib.check(lib().gank_sn_power_iter_fwd_prep(self.table, self.n, ptable, pw, len(todo), C.byref(ldesc) if ldesc is not None else None,
                                                         _stream()), "sn_power_iter_fwd_prep")
        _prep_attach(self.W_bar, kinds, todo, outs)
        if self.label is not None:
            self.W_bar[self.label[1]]._label_T = self.label_out
        return self.W_bar

    def u_out_views(self):
        out, o = [], 0
        for _, c in self.KC:
            out.append(self.u_out[o:o + c])
            o += c
        return out

    def sigma(self, i):
        return self.scal[8 * i]

    def backward(self, dW_bars, dWs):
        """dWs[i] += full SN gradient of dW_bars[i] (accumulating)."""
        for i, (g, d) in enumerate(zip(dW_bars, dWs)):
            self.table[i].dW_bar = _p(g, F32, "dW_bar").value
            self.table[i].dW = _p(d, F32, "dW").value
        _lib.check(lib().gank_sn_power_iter_bwd(self.table, self.n, _stream()), "sn_power_iter_bwd")

    def backward_gw(self, dW_bars, dWs):
        """first backward launch only (<dW_bar, W> partials); the apply step rides on the optimiser launch (sn_adam_fwd_a)"""
        for i, (g, d) in enumerate(zip(dW_bars, dWs)):
            self.table[i].dW_bar = _p(g, F32, "dW_bar").value
            self.table[i].dW = _p(d, F32, "dW").value
        _lib.check(lib().gank_sn_power_iter_bwd_gw(self.table, self.n, _stream()), "sn_power_iter_bwd_gw")

    def adam_fwd_a(self, p, g, m, v, hp, t_state, iteration=None, health=None, dw_zero=False, bump=None, bump_when_zero=None):
        """gank_sn_adam_fwd_a after backward_gw: the spectral norm's gradient, TF-Adam over the WHOLE flat buffer (p, g, m, v; the
        consumed gradients and dW_bar slices cleared) and the next forward pass's power iteration on the updated weights (u' ->
        the state's staging buffer) in one launch.  Needs the persistent state and an assigning forward pass before it.
        dw_zero: the caller guarantees the weights' own gradient views are zero (only this backward pass contributes to them).
        bump / bump_when_zero: an int64 counter advanced by one when the int32 word is 0 (the iteration count behind an iteration's last update)."""
        st = self.state
        assert st is not None and self.inplace and self.n <= 16
        ptrs = (C.c_void_p * self.n)()
        o = 0
        for i, (_, c) in enumerate(self.KC):
            ptrs[i] = st.u_next.data_ptr() + 4 * o
            o += c
        _lib.check(lib().gank_sn_adam_fwd_a(self.table, self.n, ptrs, _p(p, F32, "p"), _p(g, F32, "g"), _p(m, F32, "m"), _p(v, F32, "v"), p.numel(),
                                            _p(hp, F32, "hp"), _p(t_state, torch.int64, "t_state"), _p(iteration, torch.int64, "iteration"),
                                            _p(health, torch.int64, "health"), 1 if dw_zero else 0, _p(bump, torch.int64, "bump"),
                                            _p(bump_when_zero, I32, "bump_when_zero"), _stream()), "sn_adam_fwd_a")
        st.valid = True


# ------------------------------------------------------------------ conditional batch norm
def cbn_fwd(x, labels, gamma, beta, groups=1, relu=False, eps=1e-5):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    parts = lib().gank_cbn_parts((n // groups) * hw)
    y = torch.empty_like(x)
    stats = torch.empty((groups, 2, c), dtype=F32, device=x.device)
    ws = torch.empty(groups * parts * 3 * c, dtype=F32, device=x.device)
    _lib.check(lib().gank_cbn_fwd_eps(_p(x, BF16, "x"), _p(labels, I32, "labels"), _p(gamma, F32, "gamma"), _p(beta, F32, "beta"),
                                      _p(y), _p(stats), _p(ws), n, hw, c, groups, gamma.shape[0], int(relu), float(eps), _stream()),
               "cbn_fwd")
    return y, stats


def cbn_fwd_from_sums(x, labels, gamma, beta, cs, relu=False, eps=1e-5):
    """conditional batch norm forward on statistics a conv epilogue produced (ConvStats): one launch"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    stats = torch.empty((cs.groups, 2, c), dtype=F32, device=x.device)
    _lib.check(lib().gank_cbn_fwd_from_sums(_p(x, BF16, "x"), _p(labels, I32, "labels"), _p(gamma, F32, "gamma"), _p(beta, F32, "beta"),
                                            _p(y), _p(stats), _p(cs.sums, F32, "sums"), _p(cs.shift, F32, "shift"), n, hw, c, cs.groups,
                                            gamma.shape[0], int(relu), float(eps), _stream()), "cbn_fwd_from_sums")
    return y, stats


def cbn_stats(x, groups=1, cs=None, eps=1e-5):
    """(mean, invstd) [groups, 2, C] of x per tower: from a conv epilogue's sums (cs: ConvStats) or by the statistics pass"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    stats = torch.empty((groups, 2, c), dtype=F32, device=x.device)
    if cs is not None and cs.groups == groups and cs.sums.shape[-1] == c:
        _lib.check(lib().gank_cbn_stats_from_sums(_p(cs.sums, F32, "sums"), _p(cs.shift, F32, "shift"), _p(stats), c, groups, (n // groups) * hw,
                                                  float(eps), _stream()), "cbn_stats_from_sums")
        return stats
    parts = lib().gank_cbn_parts((n // groups) * hw)
    ws = torch.empty(groups * parts * 3 * c, dtype=F32, device=x.device)
    _lib.check(lib().gank_cbn_stats(_p(x, BF16, "x"), _p(stats), _p(ws), n, hw, c, groups, float(eps), _stream()), "cbn_stats")
    return stats


def cbn_relu_conv3x3_fprop(x, labels, gamma, beta, stats, wf, bias, cout, flags=0, out=None):
    """conv3x3_SAME(relu(cond_batchnorm(x))) + bias [tanh], the normalisation fused into the conv's operand staging (no grad).
    out: a contiguous bf16 buffer of n*h*w*cout elements to write into (any shape) -> returned as [n,h,w,cout]"""
    n, h, w, cin = x.shape
    groups = stats.shape[0]
    if out is not None:
        assert out.dtype == BF16 and out.is_contiguous() and out.numel() == n * h * w * cout and out.device == x.device
        y = out.view(n, h, w, cout)
    else:
        y = torch.empty((n, h, w, cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_cbn_relu_conv3x3_fprop(_p(x, BF16, "x"), _p(labels, I32, "labels"), _p(gamma, F32, "gamma"), _p(beta, F32, "beta"),
                                                 _p(stats, F32, "stats"), _p(wf, BF16, "wf"), _p(bias, F32, "bias"), _p(y), n, h, w, cin, cout,
                                                 groups, gamma.shape[0], flags, _stream()), "cbn_relu_conv3x3_fprop")
    return y


def layer_norm_fwd(x, gamma, beta, eps=1e-12):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    stats = torch.empty((n, 2), dtype=F32, device=x.device)
    _lib.check(lib().gank_layer_norm_fwd(_p(x, BF16, "x"), _p(gamma, F32, "gamma"), _p(beta, F32, "beta"), _p(y), _p(stats),
                                         n, hw, c, float(eps), _stream()), "layer_norm_fwd")
    return y, stats


def layer_norm_bwd(dy, x, gamma, stats, dgamma, dbeta):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    dx = torch.empty_like(x)
    _lib.check(lib().gank_layer_norm_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(gamma, F32, "gamma"), _p(stats, F32, "stats"), _p(dx),
                                         _p(dgamma, F32, "dgamma"), _p(dbeta, F32, "dbeta"), n, hw, c, _stream()), "layer_norm_bwd")
    return dx


def pixel_norm_fwd(x, eps=1e-8):
    c = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(lib().gank_pixel_norm_fwd(_p(x, BF16, "x"), _p(y), x.numel() // c, c, float(eps), _stream()), "pixel_norm_fwd")
    return y


def pixel_norm_bwd(dy, x, eps=1e-8):
    c = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(lib().gank_pixel_norm_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(dx), x.numel() // c, c, float(eps), _stream()),
               "pixel_norm_bwd")
    return dx


CBN_BWD_PART_ROWS = True


def cbn_bwd(dy, x, y, labels, gamma, stats, dgamma, dbeta, groups=1, relu=False, beta=None):
    """beta given (and relu): the mask is recomputed from x, gamma, beta and the statistics instead of read from y"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    dx = torch.empty_like(x)
    # the larger workspace: per-part rows instead of a fill launch + fp32 atomics (CBN_BWD_PART_ROWS = False: the two older entries)
    if CBN_BWD_PART_ROWS:
        nws = int(lib().gank_cbn_bwd_ws_floats(n, hw, c, groups))
        ws = torch.empty(nws, dtype=F32, device=x.device)
        remask = beta is not None and relu
        _lib.check(lib().gank_cbn_bwd_ws(_p(dy, BF16, "dy"), _p(x, BF16, "x"), None if remask else _p(y, BF16, "y"), _p(beta, F32, "beta") if remask else None,
                                         _p(labels, I32, "labels"), _p(gamma, F32, "gamma"), _p(stats, F32), _p(dx), _p(dgamma, F32, "dgamma"),
                                         _p(dbeta, F32, "dbeta"), _p(ws), nws, n, hw, c, groups, gamma.shape[0], int(relu), _stream()), "cbn_bwd_ws")
        return dx
    ws = torch.empty(n * 2 * c + groups * 2 * c, dtype=F32, device=x.device)
    if beta is not None and relu:
        _lib.check(lib().gank_cbn_bwd_remask(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(beta, F32, "beta"), _p(labels, I32, "labels"),
                                             _p(gamma, F32, "gamma"), _p(stats, F32), _p(dx), _p(dgamma, F32, "dgamma"),
                                             _p(dbeta, F32, "dbeta"), _p(ws), n, hw, c, groups, gamma.shape[0], int(relu), _stream()),
                   "cbn_bwd_remask")
        return dx
    _lib.check(lib().gank_cbn_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(y, BF16, "y"), _p(labels, I32, "labels"),
                                  _p(gamma, F32, "gamma"), _p(stats, F32), _p(dx), _p(dgamma, F32, "dgamma"),
                                  _p(dbeta, F32, "dbeta"), _p(ws), n, hw, c, groups, gamma.shape[0], int(relu), _stream()),
               "cbn_bwd")
    return dx


# ------------------------------------------------------------------ glue
def pool2x2(x, scale=0.25, residual=None):
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=BF16, device=x.device)
    _lib.check(lib().gank_pool2x2(_p(x, BF16, "x"), _p(residual, BF16, "residual"), _p(y), n, h // 2, w // 2, c, scale, _stream()), "pool2x2")
    return y


def unpool2x2_add(g, base=None, scale=0.25):
    n, h, w, c = g.shape
    y = torch.empty((n, 2 * h, 2 * w, c), dtype=BF16, device=g.device)
    _lib.check(lib().gank_unpool2x2_add(_p(g, BF16, "g"), _p(base, BF16, "base"), _p(y), n, h, w, c, scale, _stream()), "unpool2x2_add")
    return y


def add(a, b):
    y = torch.empty_like(a)
    _lib.check(lib().gank_add_bf16(_p(a, BF16, "a"), _p(b, BF16, "b"), _p(y), a.numel(), _stream()), "add")
    return y


def relu_fwd(x, leak=0.0):
    y = torch.empty_like(x)
    _lib.check(lib().gank_relu_fwd(_p(x, BF16, "x"), _p(y), x.numel(), leak, _stream()), "relu_fwd")
    return y


def relu_bwd(dy, x, leak=0.0):
    dx = torch.empty_like(x)
    _lib.check(lib().gank_relu_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(dx), x.numel(), leak, _stream()), "relu_bwd")
    return dx


def tanh_bwd(dy, y):
    dx = torch.empty_like(y)
    _lib.check(lib().gank_tanh_bwd(_p(dy, BF16, "dy"), _p(y, BF16, "y"), _p(dx), y.numel(), _stream()), "tanh_bwd")
    return dx


def copy_(dst, src):
    """dst <- src (same dtype and element count, both contiguous) as a kernel launch, never a memcpy node."""
    assert dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous(), \
        (dst.dtype, src.dtype, tuple(dst.shape), tuple(src.shape))
    _lib.check(lib().gank_copy_bytes(_p(dst, None, "dst"), _p(src, None, "src"), dst.numel() * dst.element_size(), _stream()), "copy_bytes")
    return dst


def copy_gather_(dst, srcs):
    """dst[i] <- srcs[i] (equal-sized contiguous tensors of dst's dtype, at most 16) in one kernel launch"""
    n = len(srcs)
    assert dst.is_contiguous() and dst.shape[0] == n and all(s.dtype == dst.dtype and s.is_contiguous() and s.numel() == dst[0].numel() for s in srcs), \
        (tuple(dst.shape), [tuple(s.shape) for s in srcs])
    ptrs = (C.c_void_p * n)(*[_p(s, None, "src").value for s in srcs])
    _lib.check(lib().gank_copy_bytes_gather(_p(dst, None, "dst"), ptrs, n, dst[0].numel() * dst.element_size(), _stream()), "copy_bytes_gather")
    return dst


def copy_gather2_(dst_a, srcs_a, dst_b, srcs_b):
    """copy_gather_(dst_a, srcs_a) and copy_gather_(dst_b, srcs_b) in ONE launch (gank_copy_bytes_gather2)"""
    for dst, srcs in ((dst_a, srcs_a), (dst_b, srcs_b)):
        assert dst.is_contiguous() and dst.shape[0] == len(srcs) and all(s.dtype == dst.dtype and s.is_contiguous() and s.numel() == dst[0].numel() for s in srcs), \
            (tuple(dst.shape), [tuple(s.shape) for s in srcs])
    pa = (C.c_void_p * len(srcs_a))(*[_p(s, None, "src").value for s in srcs_a])
    pb = (C.c_void_p * len(srcs_b))(*[_p(s, None, "src").value for s in srcs_b])
    _lib.check(lib().gank_copy_bytes_gather2(_p(dst_a, None, "dst"), pa, len(srcs_a), dst_a[0].numel() * dst_a.element_size(),
                                             _p(dst_b, None, "dst"), pb, len(srcs_b), dst_b[0].numel() * dst_b.element_size(), _stream()), "copy_bytes_gather2")


def clone(src):
    return copy_(torch.empty_like(src), src)


def scale_f32(x, s):
    y = torch.empty_like(x)
    _lib.check(lib().gank_scale_f32(_p(x, F32, "x"), _p(s, F32, "s"), _p(y), x.numel(), _stream()), "scale_f32")
    return y


def weighted_sum_f32(terms, weights, out=None):
    """out = sum_i weights[i] * terms[i] (fp32 tensors of one shape, at most 4; out may be one of them)"""
    assert 1 <= len(terms) <= 4 and len(terms) == len(weights)
    out = torch.empty_like(terms[0]) if out is None else out
    ts = list(terms) + [None] * (4 - len(terms))
    ws = [float(w) for w in weights] + [0.0] * (4 - len(weights))
    _lib.check(lib().gank_weighted_sum4_f32(_p(ts[0], F32, "term 0"), _p(ts[1], F32, "term 1"), _p(ts[2], F32, "term 2"), _p(ts[3], F32, "term 3"),
                                            ws[0], ws[1], ws[2], ws[3], _p(out, F32, "out"), terms[0].numel(), _stream()), "weighted_sum4_f32")
    return out


def to_bf16(x):
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    _lib.check(lib().gank_cast_f32_bf16(_p(x, F32, "x"), _p(y), x.numel(), _stream()), "cast_f32_bf16")
    return y


def to_f32(x, out=None):
    y = torch.empty(x.shape, dtype=F32, device=x.device) if out is None else out
    assert y.numel() == x.numel() and y.dtype == F32
    _lib.check(lib().gank_cast_bf16_f32(_p(x, BF16, "x"), _p(y), x.numel(), _stream()), "cast_bf16_f32")
    return y


def relu_meanpool_hw_fwd(x):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty((n, c), dtype=BF16, device=x.device)
    _lib.check(lib().gank_relu_meanpool_hw_fwd(_p(x, BF16, "x"), _p(y), n, hw, c, _stream()), "relu_meanpool_fwd")
    return y


def relu_meanpool_hw_bwd(dy, x):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    dx = torch.empty_like(x)
    _lib.check(lib().gank_relu_meanpool_hw_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(dx), n, hw, c, _stream()), "relu_meanpool_bwd")
    return dx


def concat_tile_fwd(a, e):
    n, h, w, c1 = a.shape
    c2 = e.shape[1]
    y = torch.empty((n, h, w, c1 + c2), dtype=BF16, device=a.device)
    _lib.check(lib().gank_concat_tile_fwd(_p(a, BF16, "a"), _p(e, BF16, "e"), _p(y), n, h * w, c1, c2, _stream()), "concat_tile_fwd")
    return y


def concat_tile_bwd(dy, c1):
    n, h, w, c = dy.shape
    da = torch.empty((n, h, w, c1), dtype=BF16, device=dy.device)
    de = torch.empty((n, c - c1), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_concat_tile_bwd(_p(dy, BF16, "dy"), _p(da), _p(de), n, h * w, c1, c - c1, _stream()), "concat_tile_bwd")
    return da, de


def embedding_fwd(table, idx):
    n, (vocab, d) = idx.numel(), table.shape
    y = torch.empty((n, d), dtype=BF16, device=table.device)
    _lib.check(lib().gank_embedding_fwd(_p(table, F32, "table"), _p(idx, I32, "idx"), _p(y), n, d, vocab, _stream()), "embedding_fwd")
    return y


def embedding_bwd(dy, idx, dtable):
    n, (vocab, d) = idx.numel(), dtable.shape
    _lib.check(lib().gank_embedding_bwd(_p(dy, BF16, "dy"), _p(idx, I32, "idx"), _p(dtable, F32, "dtable"), n, d, vocab, _stream()), "embedding_bwd")
    return dtable


def label_dense_table(table, w, bias=None, sigma=None):
    """T [V, Cout] bf16 = bf16(bf16(table) (w / sigma) + bias): the per-label rows of embed_y -> Linear (gan_cifar_resnet.py:276-281)"""
    v, d = table.shape
    cout = w.shape[1]
    assert w.shape[0] == d, (table.shape, w.shape)
    out = torch.empty((v, cout), dtype=BF16, device=table.device)
    _lib.check(lib().gank_label_dense_table(_p(table, F32, "table"), _p(w, F32, "w"), _p(sigma, F32, "sigma"), _p(bias, F32, "bias"), _p(out),
                                            v, d, cout, _stream()), "label_dense_table")
    return out


def concat_label_fwd(a, t, labels):
    n, h, w, c1 = a.shape
    v, c2 = t.shape
    y = torch.empty((n, h, w, c1 + c2), dtype=BF16, device=a.device)
    _lib.check(lib().gank_concat_label_fwd(_p(a, BF16, "a"), _p(t, BF16, "T"), _p(labels, I32, "labels"), _p(y), n, h * w, c1, c2, v, _stream()),
               "concat_label_fwd")
    return y


def concat_label_bwd(dy, c1):
    """-> (da bf16 [N,H,W,C1], de32 fp32 [N, C2])"""
    n, h, w, c = dy.shape
    da = torch.empty((n, h, w, c1), dtype=BF16, device=dy.device)
    de = torch.empty((n, c - c1), dtype=F32, device=dy.device)
    _lib.check(lib().gank_concat_label_bwd(_p(dy, BF16, "dy"), _p(da), _p(de), n, h * w, c1, c - c1, _stream()), "concat_label_bwd")
    return da, de


def concat_label_pool_fwd(a, t, labels, want_full=True):
    """-> (y [N,H,W,C1+C2], mean_pool2x2(y)) in one pass; want_full=False: (None, mean_pool2x2(y)) -- the full tensor is not written"""
    n, h, w, c1 = a.shape
    v, c2 = t.shape
    y = torch.empty((n, h, w, c1 + c2), dtype=BF16, device=a.device) if want_full else None
    yp = torch.empty((n, h // 2, w // 2, c1 + c2), dtype=BF16, device=a.device)
    _lib.check(lib().gank_concat_label_pool_fwd(_p(a, BF16, "a"), _p(t, BF16, "T"), _p(labels, I32, "labels"), _p(y), _p(yp), n, h, w, c1, c2, v, _stream()),
               "concat_label_pool_fwd")
    return y, yp


def concat_label_unpool_bwd_factored(g_main_c1, g_pooled, de_add=None, labels=None, lists=None):
    """concat_label_unpool_bwd where the consumer of the tiled half was factored out: g_main_c1 [N,H,W,C1] (the first C1 channels'
    gradient), de_add fp32 [parts,V,C2] that consumer's gradient of the tiled vector per LABEL (added to the row of the label's first
    sample: labels, lists from label_conv3x3_table) -> (da bf16 [N,H,W,C1], de32 fp32 [N,C2])"""
    n, hp, wp, c = g_pooled.shape
    want_da = not isinstance(g_main_c1, int)          # an int = C1: only the tiled vector's gradient (the feature join happened elsewhere)
    c1 = g_main_c1.shape[3] if want_da else int(g_main_c1)
    da = torch.empty((n, 2 * hp, 2 * wp, c1), dtype=BF16, device=g_pooled.device) if want_da else None
    if not want_da:
        g_main_c1 = None
    de = torch.empty((n, c - c1), dtype=F32, device=g_pooled.device)
    parts = 0 if de_add is None else de_add.shape[0]
    v = 0 if de_add is None else de_add.shape[1]
    _lib.check(lib().gank_concat_label_unpool_bwd_factored(_p(g_main_c1, BF16, "g_main_c1"), _p(g_pooled, BF16, "g_pooled"), _p(da), _p(de),
                                                           _p(de_add, F32, "de_add"), parts, _p(labels, I32, "labels"), _p(lists, I32, "lists"), v,
                                                           n, 2 * hp, 2 * wp, c1, c - c1, _stream()),
               "concat_label_unpool_bwd_factored")
    return da, de


# ---- the spatially constant input channels of a 3x3 conv, factored out (csrc/label_conv.hip)
def label_conv3x3_table(w, c0, t, bias=None, labels=None):
    """bias_table fp32 [V,9,Cout] = bias + what the constant channels c0.. of the fp32 filter w [3,3,Cin,Cout] contribute per (label,
    border class) with the per-label vectors relu(t[v]) (t bf16 [V,C2]).  labels given: -> (bias_table, lists int32 [V,N+1]: row v =
    {count, the samples of label v in ascending order} for the backward entries)"""
    v, c2 = t.shape
    cin, cout = w.shape[2], w.shape[3]
    out = torch.empty((v, 9, cout), dtype=F32, device=w.device)
    n = 0 if labels is None else labels.numel()
    lists = torch.empty((v, n + 1), dtype=I32, device=w.device) if labels is not None else None
    _lib.check(lib().gank_label_conv3x3_table(_p(w, F32, "w"), cin, c0, c2, cout, _p(t, BF16, "T"), v, _p(bias, F32, "bias"), _p(out),
                                              _p(labels, I32, "labels"), n, _p(lists), _stream()), "label_conv3x3_table")
    return out if labels is None else (out, lists)


def label_conv3x3_table_pooled_shortcut_ok(a, c2, ws_f, cs):
    """the one-workgroup-per-sample form: 64 pooled pixels, 128 shortcut channels, at most 256 input channels"""
    n, h, wd_, c1 = a.shape
    return ((h // 2) * (wd_ // 2) == 64 and h % 2 == 0 and wd_ % 2 == 0 and cs == 128 and (c1 + c2) % 16 == 0 and c1 + c2 <= 256
            and c1 % 8 == 0 and c1 <= 128 and c2 % 8 == 0 and 576 % (c2 // 8) == 0
            and ws_f is not None and ws_f.dim() == 2 and ws_f.shape[0] >= cs and ws_f.shape[1] >= c1 + c2 and ws_f.shape[1] % 8 == 0)


def label_conv3x3_table_pooled(w, c0, t, bias, labels, a, shortcut=None):
    """label_conv3x3_table(labels=...) and concat_label_pool_fwd(a, t, labels, want_full=False) in ONE launch
    -> (bias_table, lists, mean_pool2x2(concat(a, tile(t[labels]))))
    shortcut = (ws_f bf16 [Cs, >= C1 + C2] (the plain-conv operand of a 1x1 filter), bias_s | None, Cs): the launch also computes
    conv1x1(pooled) + bias_s -> a fourth result [N, H/2, W/2, Cs] (gank_label_conv3x3_table_pooled_shortcut)"""
    v, c2 = t.shape
    cin, cout = w.shape[2], w.shape[3]
    n, h, wd_, c1 = a.shape
    out = torch.empty((v, 9, cout), dtype=F32, device=w.device)
    lists = torch.empty((v, n + 1), dtype=I32, device=w.device)
    yp = torch.empty((n, h // 2, wd_ // 2, c1 + c2), dtype=BF16, device=a.device)
    if shortcut is not None:
        ws_f, bias_s, cs = shortcut
        sc = torch.empty((n, h // 2, wd_ // 2, cs), dtype=BF16, device=a.device)
        _lib.check(lib().gank_label_conv3x3_table_pooled_shortcut(_p(w, F32, "w"), cin, c0, c2, cout, _p(t, BF16, "T"), v, _p(bias, F32, "bias"), _p(out),
                                                                  _p(labels, I32, "labels"), n, _p(lists), _p(a, BF16, "a"), _p(yp), h, wd_, c1,
                                                                  _p(ws_f, BF16, "ws_f"), ws_f.shape[1], _p(bias_s, F32, "bias_s"), cs, _p(sc), _stream()),
                   "label_conv3x3_table_pooled_shortcut")
        return out, lists, yp, sc
    _lib.check(lib().gank_label_conv3x3_table_pooled(_p(w, F32, "w"), cin, c0, c2, cout, _p(t, BF16, "T"), v, _p(bias, F32, "bias"), _p(out),
                                                     _p(labels, I32, "labels"), n, _p(lists), _p(a, BF16, "a"), _p(yp), h, wd_, c1, _stream()),
               "label_conv3x3_table_pooled")
    return out, lists, yp


def img16_conv3x3_label_bias(x, rf, bias_table, labels, cout, flags=0):
    """gank_img16_conv3x3_label_bias: the image-resident conv on the feature channels x [N,16,16,Cin] (rf: kind-6 operand) plus row
    (label, border class) of the table"""
    n, cin = x.shape[0], x.shape[3]
    assert tuple(x.shape[1:3]) == (16, 16) and bias_table.shape[1:] == (9, cout), (x.shape, bias_table.shape)
    y = torch.empty((n, 16, 16, cout), dtype=BF16, device=x.device)
    _lib.check(lib().gank_img16_conv3x3_label_bias(_p(x, BF16, "x"), _p(rf, BF16, "rf"), _p(bias_table, F32, "bias_table"), _p(labels, I32, "labels"),
                                                   bias_table.shape[0], _p(y), n, cin, cout, flags, _stream()), "img16_conv3x3_label_bias")
    return y


def img16_conv3x3_dgrad_unpool(dy, rd, relu_ref, g_pooled, cout, scale=0.25):
    """gank_img16_conv3x3_dgrad_unpool: relu_mask(conv(dy)) + scale * unpool2x(g_pooled[..., :cout]) -> [N,16,16,cout]"""
    n, cin = dy.shape[0], dy.shape[3]
    assert tuple(dy.shape[1:3]) == (16, 16) and tuple(g_pooled.shape[:3]) == (n, 8, 8) and g_pooled.shape[3] >= cout
    y = torch.empty((n, 16, 16, cout), dtype=BF16, device=dy.device)
    _lib.check(lib().gank_img16_conv3x3_dgrad_unpool(_p(dy, BF16, "dy"), _p(rd, BF16, "rd"), _p(relu_ref, BF16, "relu_ref"), _p(g_pooled, BF16, "g_pooled"),
                                                     g_pooled.shape[3], float(scale), _p(y), n, cin, cout, _stream()), "img16_conv3x3_dgrad_unpool")
    return y


def img16_conv3x3_label_bwd(dy, rd, relu_ref, cin_out, sums, t, w, c0, dw):
    """gank_img16_conv3x3_label_bwd: the factored layer's input gradient (dy [N,16,16,Cout] -> [N,16,16,cin_out], masked by relu_ref) and,
    as extra workgroups of the same launch, label_conv3x3_bwd's label gradients on the tap sums `sums` -> (dx, de_parts fp32 [9,V,C2])"""
    n, cout = dy.shape[0], dy.shape[3]
    v, c2 = t.shape
    dx = torch.empty((n, 16, 16, cin_out), dtype=BF16, device=dy.device)
    parts = torch.empty((9, v, c2), dtype=F32, device=dy.device)
    _lib.check(lib().gank_img16_conv3x3_label_bwd(_p(dy, BF16, "dy"), _p(rd, BF16, "rd"), _p(relu_ref, BF16, "relu_ref"), _p(dx), n, cout, cin_out, 0,
                                                  _p(sums, F32, "sums"), _p(t, BF16, "T"), v, _p(w, F32, "w"), w.shape[2], c0, c2, w.shape[3],
                                                  _p(dw, F32, "dw"), _p(parts), _stream()), "img16_conv3x3_label_bwd")
    return dx, parts


def label_conv3x3_bwd(dy, lists, t, w, c0, dw, dw_feat_tmp=None, sums=None):
    """gank_label_conv3x3_bwd: ACCUMULATES the constant channels' filter gradient into rows c0.. of dw [3,3,Cin,Cout] (and adds +
    clears dw_feat_tmp [3,3,c0,Cout] into rows 0..c0-1) -> de_parts fp32 [9,V,C2]: the gradient of the tiled vector per tap and LABEL
    (summed over the label's samples); lists from label_conv3x3_table(labels=...).  sums: the tap sums of dy are in this workspace
    already (conv2d_wgrad_rows(tap_sums=...)): the launch that computes them is skipped"""
    n, h, wd_, cout = dy.shape
    v, c2 = t.shape
    cin = w.shape[2]
    assert dw.shape == w.shape and (dw_feat_tmp is None or tuple(dw_feat_tmp.shape) == (3, 3, c0, cout))
    ws = sums if sums is not None else torch.empty(int(lib().gank_label_conv3x3_bwd_ws_floats(n, cout)), dtype=F32, device=dy.device)
    parts = torch.empty((9, v, c2), dtype=F32, device=dy.device)
    _lib.check(lib().gank_label_conv3x3_bwd(None if sums is not None else _p(dy, BF16, "dy"), _p(lists, I32, "lists"), _p(t, BF16, "T"), v, _p(w, F32, "w"), cin, c0, c2, cout, n, h, wd_,
                                            _p(dw, F32, "dw"), _p(dw_feat_tmp, F32, "dw_feat_tmp"), _p(parts), _p(ws), _stream()), "label_conv3x3_bwd")
    return parts


def label_conv3x3_bwd_pooled(sums, lists, t, w, c0, dw, g_pooled, c0g, n, dw_feat_tmp=None):
    """gank_label_conv3x3_bwd_pooled: label_conv3x3_bwd(sums=...) -> de_parts fp32 [10,V,C2] whose tenth part is the per-label sum of
    g_pooled[..., c0g:c0g+C2] (the pooled shortcut branch's gradient of the tiled vector) from extra workgroups of the same launch"""
    v, c2 = t.shape
    cin, cout = w.shape[2], w.shape[3]
    assert dw.shape == w.shape and g_pooled.shape[0] == n
    hwp, pitch = g_pooled.shape[1] * g_pooled.shape[2], g_pooled.shape[3]
    parts = torch.empty((10, v, c2), dtype=F32, device=g_pooled.device)
    _lib.check(lib().gank_label_conv3x3_bwd_pooled(_p(sums, F32, "sums"), _p(lists, I32, "lists"), _p(t, BF16, "T"), v, _p(w, F32, "w"), cin, c0, c2, cout, n,
                                                   _p(dw, F32, "dw"), _p(dw_feat_tmp, F32, "dw_feat_tmp"), _p(parts), _p(g_pooled, BF16, "g_pooled"), hwp, pitch, c0g,
                                                   _stream()), "label_conv3x3_bwd_pooled")
    return parts


def label_dense_bwd_parts(parts, table, w, dw=None, dbias=None, dtable=None):
    """label_dense_bwd from rows summed per label already: parts fp32 [P,V,C2], dT[l] = sum_p parts[p][l]"""
    npart, v, c2 = parts.shape
    assert table.shape[0] == v
    _lib.check(lib().gank_label_dense_bwd_parts(_p(parts, F32, "parts"), npart, _p(table, F32, "table"), _p(w, F32, "w"), _p(dw, F32, "dw"),
                                                _p(dbias, F32, "dbias"), _p(dtable, F32, "dtable"), v, table.shape[1], c2, _stream()), "label_dense_bwd_parts")


def concat_label_unpool_bwd(g_main, g_pooled, c1):
    """gradients of concat_label_pool_fwd's two outputs -> (da bf16 [N,H,W,C1], de32 fp32 [N,C2])"""
    n, hp, wp, c = g_pooled.shape
    da = torch.empty((n, 2 * hp, 2 * wp, c1), dtype=BF16, device=g_pooled.device)
    de = torch.empty((n, c - c1), dtype=F32, device=g_pooled.device)
    _lib.check(lib().gank_concat_label_unpool_bwd(_p(g_main, BF16, "g_main"), _p(g_pooled, BF16, "g_pooled"), _p(da), _p(de), n, 2 * hp, 2 * wp, c1, c - c1,
                                                  _stream()), "concat_label_unpool_bwd")
    return da, de


def label_dense_bwd(de32, labels, table, w, dw=None, dbias=None, dtable=None):
    """ACCUMULATES dw [D,C2] += bf16(table)^T dT, dbias [C2] += sum_l dT[l], dtable [V,D] += dT w^T with dT[l] = sum of de32 rows of label l"""
    n, c2 = de32.shape
    v, d = table.shape
    _lib.check(lib().gank_label_dense_bwd(_p(de32, F32, "de32"), _p(labels, I32, "labels"), _p(table, F32, "table"), _p(w, F32, "w"),
                                          _p(dw, F32, "dw"), _p(dbias, F32, "dbias"), _p(dtable, F32, "dtable"), n, v, d, c2, _stream()),
               "label_dense_bwd")


# ------------------------------------------------------------------ losses / optimiser / input
def hinge_d_loss(logits, n_real, loss=None):
    """-> (loss fp32[1], dlogits bf16, dlogits fp32): the bf16 copy is the gradient for an upstream gradient of 1.
    `loss`: a persistent fp32[1] buffer to write the value into (the trainers' reported loss: no copy launch)"""
    loss = torch.empty(1, dtype=F32, device=logits.device) if loss is None else loss
    dl = torch.empty_like(logits)
    dl32 = torch.empty(logits.shape, dtype=F32, device=logits.device)
    _lib.check(lib().gank_hinge_d_loss(_p(logits, BF16, "logits"), _p(loss), _p(dl), _p(dl32), logits.numel(), n_real, _stream()), "hinge_d_loss")
    return loss, dl, dl32


def wgan_d_loss(logits, n_real, loss=None):
    loss = torch.empty(1, dtype=F32, device=logits.device) if loss is None else loss
    dl = torch.empty_like(logits)
    dl32 = torch.empty(logits.shape, dtype=F32, device=logits.device)
    _lib.check(lib().gank_wgan_d_loss(_p(logits, BF16, "logits"), _p(loss), _p(dl), _p(dl32), logits.numel(), n_real, _stream()), "wgan_d_loss")
    return loss, dl, dl32


def gan_pointwise_loss(logits, n_real, kind, loss=None):
    """kind 0 LSGAN critic | 1 LSGAN generator | 2 sigmoid-xent critic | 3 -log sigmoid generator | 4 MiniMax generator |
    5 / 6 / 7 the SOFT_PLUS 'Goodfellow' critic / generator and 'HINGE' critic of SNGAN/gan_cifar_resnet.py:364-386
    (gank_gan_pointwise_loss) -> (loss fp32[1], dlogits bf16, dlogits fp32)"""
    loss = torch.empty(1, dtype=F32, device=logits.device) if loss is None else loss
    dl = torch.empty_like(logits)
    dl32 = torch.empty(logits.shape, dtype=F32, device=logits.device)
    _lib.check(lib().gank_gan_pointwise_loss(_p(logits, BF16, "logits"), _p(loss), _p(dl), _p(dl32), logits.numel(), int(n_real), int(kind), _stream()),
               "gan_pointwise_loss")
    return loss, dl, dl32


def hinge_g_loss(logits, loss=None):
    loss = torch.empty(1, dtype=F32, device=logits.device) if loss is None else loss
    dl = torch.empty_like(logits)
    dl32 = torch.empty(logits.shape, dtype=F32, device=logits.device)
    _lib.check(lib().gank_hinge_g_loss(_p(logits, BF16, "logits"), _p(loss), _p(dl), _p(dl32), logits.numel(), _stream()), "hinge_g_loss")
    return loss, dl, dl32


def critic_head_hinge(x, w, b, n_real, mode, want_dx=True, w_grad=None, b_grad=None, loss=None, loss_scale=1.0):
    """fused D.Output + hinge loss (gank_critic_head_hinge) -> (loss fp32[1], logits bf16 [M], dx bf16 [M,K] | None);
    w_grad / b_grad are ACCUMULATED when given; loss_scale (a power of two) multiplies the three gradients, not the loss"""
    m, k = x.shape
    assert w.numel() == k and (b is None or b.numel() == 1)
    loss = torch.empty(1, dtype=F32, device=x.device) if loss is None else loss
    logits = torch.empty(m, dtype=BF16, device=x.device)
    dx = torch.empty((m, k), dtype=BF16, device=x.device) if want_dx else None
    _lib.check(lib().gank_critic_head_hinge_scaled(_p(x, BF16, "x"), _p(w, F32, "w"), _p(b, F32, "b"), _p(logits), _p(loss), _p(dx), _p(w_grad, F32, "w_grad"),
                                                   _p(b_grad, F32, "b_grad"), m, k, int(n_real), int(mode), float(loss_scale), _stream()), "critic_head_hinge")
    return loss, logits, dx


def softmax_xent(logits, labels):
    n, classes = logits.shape
    loss = torch.empty(1, dtype=F32, device=logits.device)
    dl = torch.empty_like(logits)
    dl32 = torch.empty(logits.shape, dtype=F32, device=logits.device)
    _lib.check(lib().gank_softmax_xent(_p(logits, BF16, "logits"), _p(labels, I32, "labels"), _p(loss), _p(dl), _p(dl32), n, classes, _stream()), "softmax_xent")
    return loss, dl, dl32


def loss_grad_scale(dl32, g):
    """bf16(g[0] * dl32): the gradient of a loss that entered a weighted sum (g = fp32[1] upstream gradient)"""
    out = torch.empty(dl32.shape, dtype=BF16, device=dl32.device)
    _lib.check(lib().gank_loss_grad_scale(_p(dl32, F32, "dl32"), _p(g, F32, "g"), _p(out), dl32.numel(), _stream()), "loss_grad_scale")
    return out


def adam_tf(p, g, m, v, hp, t_state, iteration=None, zero_grads=False, health=None):
    """hp fp32[8] = {lr, beta1, beta2, eps, grad_scale, decay_on, 0, 0}; t_state int64[1]; iteration int64[1] | None.
    zero_grads: clear g (ALL of it: it may be longer than p, e.g. a scratch half behind the gradients) in the same launch.
    health: int64[2] counters (non-finite gradients, zero gradients), accumulated."""
    assert g.numel() >= p.numel() and hp.numel() >= 8
    _lib.check(lib().gank_adam_tf_health(_p(p, F32, "p"), _p(g, F32, "g"), _p(m, F32, "m"), _p(v, F32, "v"), _p(hp, F32, "hp"),
                                         _p(t_state, torch.int64, "t_state"), _p(iteration, torch.int64, "iteration"),
                                         p.numel(), g.numel() if zero_grads else 0, _p(health, torch.int64, "health"), _stream()), "adam_tf")


def counter_add(counter, inc=1):
    _lib.check(lib().gank_counter_add(_p(counter, torch.int64, "counter"), int(inc), _stream()), "counter_add")


def new_rng_state(seed, device):
    """{seed, offset} uint64[2] on the device (stored as int64 bits)."""
    return torch.tensor([int(seed) & ((1 << 63) - 1), 0], dtype=torch.int64, device=device)


def preprocess_real(data_u8, rng_state):
    b = data_u8.shape[0]
    assert data_u8.dtype == torch.uint8 and data_u8.shape[1] == 3072
    y = torch.empty((b, 32, 32, 3), dtype=BF16, device=data_u8.device)
    _lib.check(lib().gank_preprocess_real(_p(data_u8, torch.uint8, "data"), _p(y), _p(rng_state, torch.int64), b, _stream()), "preprocess_real")
    return y


def rng_normal(shape, rng_state):
    y = torch.empty(shape, dtype=BF16, device=rng_state.device)
    _lib.check(lib().gank_rng_normal_bf16(_p(y), y.numel(), _p(rng_state, torch.int64), _stream()), "rng_normal")
    return y


def generator_feed(rng_state, noise_shape, zero_floats=0, n_lab=0, n_labels=10):
    """gank_generator_feed: (labels int32 [n_lab] | None, noise bf16 [noise_shape], zeroed fp32 [zero_floats] | None) from ONE launch --
    rng_labels, rng_normal and a zero fill with the values and the stream offset of the separate calls"""
    dev = rng_state.device
    labels = torch.empty(n_lab, dtype=I32, device=dev) if n_lab else None
    y = torch.empty(noise_shape, dtype=BF16, device=dev)
    zb = torch.empty(int(zero_floats), dtype=F32, device=dev) if zero_floats else None
    _lib.check(lib().gank_generator_feed(_p(labels), n_lab, n_labels, _p(y), y.numel(), _p(zb), int(zero_floats), _p(rng_state, torch.int64), _stream()),
               "generator_feed")
    return labels, y, zb


def rng_labels(n, n_labels, rng_state):
    y = torch.empty(n, dtype=I32, device=rng_state.device)
    _lib.check(lib().gank_rng_labels(_p(y), n, n_labels, _p(rng_state, torch.int64), _stream()), "rng_labels")
    return y


# ------------------------------------------------------------------ ACGAN configuration
def bn_bwd_bwd(ggI, dy, x, gamma, stats, gG=None):
    """second-order train-mode batch norm (gank_bn_bwd_bwd) -> (gI, ggO); gG (fp32 [C]) accumulated when given"""
    c = x.shape[-1]
    rows = x.numel() // c
    gI, ggO = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(5 * c, dtype=F32, device=x.device)
    _lib.check(lib().gank_bn_bwd_bwd(_p(ggI, BF16, "ggI"), _p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(gamma, F32, "gamma"), _p(stats, F32, "stats"),
                                     _p(gI), _p(ggO), _p(gG, F32, "gG"), _p(ws), rows, c, _stream()), "bn_bwd_bwd")
    return gI, ggO


def bn_moving_update(stats, mm, mv, biased, step, count, decay=0.9, eps=1e-5):
    groups, _, c = stats.shape
    _lib.check(lib().gank_bn_moving_update(_p(stats, F32, "stats"), _p(mm, F32, "moving_mean"), _p(mv, F32, "moving_variance"), _p(biased, F32, "biased"),
                                           _p(step, F32, "local_step"), c, groups, int(count), float(decay), float(eps), _stream()), "bn_moving_update")


def gp_loss(grad, lam=10.0):
    """grad bf16 [N, ...] -> (loss fp32[1], d loss / d grad fp32)"""
    n = grad.shape[0]
    d = grad.numel() // n
    loss = torch.empty(1, dtype=F32, device=grad.device)
    dg = torch.empty(grad.shape, dtype=F32, device=grad.device)
    ws = torch.empty(n, dtype=F32, device=grad.device)
    _lib.check(lib().gank_gp_loss(_p(grad, BF16, "grad"), _p(loss), _p(dg), _p(ws), n, d, float(lam), _stream()), "gp_loss")
    return loss, dg


def lerp_rows(real, fake, alpha):
    n = real.shape[0]
    out = torch.empty_like(real)
    _lib.check(lib().gank_lerp_rows(_p(real, BF16, "real"), _p(fake, BF16, "fake"), _p(alpha, F32, "alpha"), _p(out), n, real.numel() // n, _stream()), "lerp_rows")
    return out


def sum_hw(x, scale):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty((n, c), dtype=BF16, device=x.device)
    _lib.check(lib().gank_sum_hw(_p(x, BF16, "x"), _p(y), n, hw, c, float(scale), _stream()), "sum_hw")
    return y


def bcast_hw(g, hw_shape, scale):
    n, c = g.shape
    h, w = hw_shape
    y = torch.empty((n, h, w, c), dtype=BF16, device=g.device)
    _lib.check(lib().gank_bcast_hw(_p(g, BF16, "g"), _p(y), n, h * w, c, float(scale), _stream()), "bcast_hw")
    return y


def rng_uniform(n, rng_state):
    y = torch.empty(n, dtype=F32, device=rng_state.device)
    _lib.check(lib().gank_rng_uniform_f32(_p(y), n, _p(rng_state, torch.int64), _stream()), "rng_uniform")
    return y


# ------------------------------------------------------------------ PGGAN / Pix2Pix operators
def axpby(a, b, alpha, beta=0.0):
    """alpha * a + beta * b (b may be None), bf16"""
    y = torch.empty_like(a)
    _lib.check(lib().gank_axpby_bf16(_p(a, BF16, "a"), _p(b, BF16, "b"), float(alpha), float(beta), _p(y), a.numel(), _stream()), "axpby")
    return y


def blend_dev(a, b, alpha, mode=0):
    """fade-in blend with the weight `alpha` (fp32[1]) in device memory: mode 0 (1-alpha) a + alpha b; 1 (1-alpha) a; 2 alpha a"""
    y = torch.empty_like(a)
    _lib.check(lib().gank_blend_dev(_p(a, BF16, "a"), _p(b, BF16, "b"), _p(alpha, F32, "alpha"), _p(y), a.numel(), int(mode), _stream()), "blend_dev")
    return y


def minibatch_std_fwd(x):
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    y = torch.empty(tuple(x.shape[:-1]) + (c + 1,), dtype=BF16, device=x.device)
    ws = torch.empty(hw * c + 2, dtype=F32, device=x.device)
    _lib.check(lib().gank_minibatch_std_fwd(_p(x, BF16, "x"), _p(y), _p(ws), b, hw, c, _stream()), "minibatch_std_fwd")
    return y, ws


def minibatch_std_bwd(dy, x, ws):
    b, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (b * c)
    dx = torch.empty_like(x)
    _lib.check(lib().gank_minibatch_std_bwd(_p(dy, BF16, "dy"), _p(x, BF16, "x"), _p(ws, F32, "ws"), _p(dx), b, hw, c, _stream()), "minibatch_std_bwd")
    return dx


def resize_bilinear(x, out_hw):
    n, hi, wi, c = x.shape
    y = torch.empty((n, out_hw[0], out_hw[1], c), dtype=BF16, device=x.device)
    _lib.check(lib().gank_resize_bilinear(_p(x, BF16, "x"), _p(y), n, hi, wi, out_hw[0], out_hw[1], c, _stream()), "resize_bilinear")
    return y


def pool2d(x, k, stride, pad, out_hw, mode, out=None, c_off=0):
    """mode 'max' | 'avg' (average over the in-image elements); writes channels [c_off, c_off + C) of `out` when given"""
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty((n, out_hw[0], out_hw[1], c), dtype=BF16, device=x.device)
    assert tuple(out.shape[:3]) == (n, out_hw[0], out_hw[1])
    _lib.check(lib().gank_pool2d(_p(x, BF16, "x"), _p(out, BF16, "out"), n, h, w, c, out_hw[0], out_hw[1], k, stride, pad, 0 if mode == 'max' else 1,
                                 out.shape[3], c_off, _stream()), "pool2d")
    return out


def relu_to_channels(x, out=None, c_off=0):
    """out[..., c_off : c_off + C] = relu(x); a fresh tensor of x's shape when out is None"""
    c = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib().gank_relu_to_channels(_p(x, BF16, "x"), _p(out, BF16, "out"), x.numel() // c, c, out.shape[-1], c_off, _stream()), "relu_to_channels")
    return out


def concat_channels(a, b):
    ca, cb = a.shape[-1], b.shape[-1]
    assert a.shape[:-1] == b.shape[:-1], (a.shape, b.shape)
    y = torch.empty(tuple(a.shape[:-1]) + (ca + cb,), dtype=BF16, device=a.device)
    _lib.check(lib().gank_concat_channels(_p(a, BF16, "a"), _p(b, BF16, "b"), _p(y), a.numel() // ca, ca, cb, _stream()), "concat_channels")
    return y


def split_channels(y, ca):
    c = y.shape[-1]
    a = torch.empty(tuple(y.shape[:-1]) + (ca,), dtype=BF16, device=y.device)
    b = torch.empty(tuple(y.shape[:-1]) + (c - ca,), dtype=BF16, device=y.device)
    _lib.check(lib().gank_split_channels(_p(y, BF16, "y"), _p(a), _p(b), y.numel() // c, ca, c - ca, _stream()), "split_channels")
    return a, b


def l1_loss(a, b):
    """-> (mean |a - b| fp32[1], d/da fp32)"""
    loss = torch.empty(1, dtype=F32, device=a.device)
    dl32 = torch.empty(a.shape, dtype=F32, device=a.device)
    ws = torch.empty(1024, dtype=F32, device=a.device)
    _lib.check(lib().gank_l1_loss(_p(a, BF16, "a"), _p(b, BF16, "b"), _p(loss), _p(dl32), _p(ws), a.numel(), _stream()), "l1_loss")
    return loss, dl32


def dropout_fwd(x, keep, rng_state):
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _lib.check(lib().gank_dropout_fwd(_p(x, BF16, "x"), _p(y), _p(mask), x.numel(), float(keep), _p(rng_state, torch.int64), _stream()), "dropout_fwd")
    return y, mask


def dropout_bwd(dy, mask, keep):
    dx = torch.empty_like(dy)
    _lib.check(lib().gank_dropout_bwd(_p(dy, BF16, "dy"), _p(mask, torch.uint8, "mask"), _p(dx), dy.numel(), float(keep), _stream()), "dropout_bwd")
    return dx


_deferred_feed = None


def defer_critic_feed(*args):
    """The critic's feed of the update that is about to start (arguments of critic_feed), to be launched by the FIRST spectral-norm
    forward pass that follows: as a block range of its second launch when that launch runs alone (the power iteration came with
    the previous optimiser step), else as the feed's own launch in front of it.  The caller checks deferred_critic_feed_pending()
    after building the forward pass: a pass without a spectral-norm batch would leave the feed unlaunched."""
    global _deferred_feed
    assert _deferred_feed is None, "a deferred critic feed is still pending"
    _deferred_feed = args


def take_deferred_critic_feed():
    global _deferred_feed
    f, _deferred_feed = _deferred_feed, None
    return f


def deferred_critic_feed_pending():
    return _deferred_feed is not None


def _critic_feed_desc(real_all, labels_all, fake_all, both, labels2, slot, rng_state, done):
    n_slots, b = labels_all.shape
    assert both.shape[0] == 2 * b and labels2.numel() == 2 * b and real_all.shape[0] == n_slots and fake_all.shape[0] == n_slots
    return _lib.CriticFeedDesc(_p(real_all, torch.uint8, "real_all").value, _p(labels_all, I32, "labels_all").value, _p(fake_all, BF16, "fake_all").value,
                               _p(both, BF16, "both").value, _p(labels2, I32, "labels2").value, _p(slot, I32, "slot").value,
                               _p(rng_state, torch.int64, "rng_state").value, _p(done, I32, "done").value, b, n_slots)


def critic_feed(real_all, labels_all, fake_all, both, labels2, slot, rng_state, done):
    """both/labels2 <- slot `slot[0]` of the feed ring (preprocessed reals, kept fakes, labels twice); advances slot and RNG"""
    n_slots, b = labels_all.shape
    assert both.shape[0] == 2 * b and labels2.numel() == 2 * b and real_all.shape[0] == n_slots and fake_all.shape[0] == n_slots
    _lib.check(lib().gank_critic_feed(_p(real_all, torch.uint8, "real_all"), _p(labels_all, I32, "labels_all"), _p(fake_all, BF16, "fake_all"),
                                      _p(both, BF16, "both"), _p(labels2, I32, "labels2"), _p(slot, I32, "slot"),
                                      _p(rng_state, torch.int64, "rng_state"), _p(done, I32, "done"), b, n_slots, _stream()), "critic_feed")


def tr_probe(device):
    out = torch.empty(256, dtype=I32, device=device)
    _lib.check(lib().gank_debug_tr_probe(_p(out), _stream()), "tr_probe")
    return out


# ------------------------------------------------------------------ profiler
def prof_enable(on):
    lib().gank_prof_enable(int(on))


def prof_reset():
    lib().gank_prof_reset()


def prof_collect(family):
    ms, fl = C.c_double(0), C.c_double(0)
    n = lib().gank_prof_collect(family, C.byref(ms), C.byref(fl))
    return n, ms.value, fl.value


def prof_kernels(family):
    """[(kernel symbol, launches, total_ms, flops, algorithmic_bytes)] of a family, longest total time first"""
    out, i = [], 0
    while True:
        name = C.create_string_buffer(128)
        n, ms, fl, by = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
        if not lib().gank_prof_kernel_stats(family, i, name, 128, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)):
            return out
        out.append((name.value.decode(), n.value, ms.value, fl.value, by.value))
        i += 1


def prof_bytes(family):
    return float(lib().gank_prof_bytes(int(family)))


def prof_calibrate(n=200):
    """average ms an event pair around an empty kernel reads (the fixed cost inside every profiler record)"""
    return float(lib().gank_prof_calibrate(int(n), _stream()))
