"""Spectral normalisation -- drop-in for common/ops/sn.py of the reference.

`spectral_normed_weight(W, u, num_iters=1, update_collection=None, with_sigma=False)` keeps the
reference signature (sn.py:15).  One power-iteration step, sigma = v W u'^T, W/sigma, gradient
THROUGH the iteration (sn.py:34-61 has no stop_gradient).  `u` write policy (sn.py:48-65):
update_collection=None -> u is overwritten on every execution; NO_OPS -> never; any other value
-> the pending update is appended to that collection (a list) as a callable.

MI355X-first: `precomputed(...)` runs ONE batched launch group for all spectrally normalised
weights of a network (12 in the SNGAN critic) instead of 12 x 3 dependent GEMVs.
"""
import contextlib
import warnings

import torch

from ... import functional as Fn
from ... import kernels as K
from ...store import get_default_store

NO_OPS = 'NO_OPS'

_active = []  # stack of {id(W): (W_bar, sigma)} dicts filled by `precomputed`
_grad_scratch = [None]   # a pre-ZEROED fp32 buffer the next `precomputed` may carve its dW_bar slices from


@contextlib.contextmanager
def grad_scratch(buf):
    """`buf` (zeroed by the caller for this backward pass, e.g. the scratch half of ParamStore.flatten's gradient
    buffer) replaces the per-step torch.zeros of the normalised weights' gradient slices."""
    _grad_scratch[0] = buf
    try:
        yield
    finally:
        _grad_scratch[0] = None


def _apply_update(us, batch, update_collection):
    if update_collection == NO_OPS:
        return
    new = batch.u_out_views()

    def assign():
        with torch.no_grad():
            for u, un in zip(us, new):
                K.copy_(u.view(-1), un)
    if update_collection is None:
        assign()                       # sn.py:55-56: u.assign(u_final) as a control dependency
    else:
        update_collection.append(assign)   # sn.py:64-65


def spectral_normed_weight(W, u=None, num_iters=1, update_collection=None, with_sigma=False, reuse=False):
    if num_iters != 1:
        raise NotImplementedError('the hot path uses num_iters=1 (sn.py:15 default)')
    for table in reversed(_active):
        hit = table.get(id(W))
        if hit is not None:
            return hit if with_sigma else hit[0]
    store = get_default_store()
    with store.variable_scope('spectral_norm'):
        if u is None:
            c = W.shape[-1]
            u = store.get_variable('u', [1, c], lambda rng: _trunc_normal(rng, (1, c)), trainable=False)
    if update_collection is None:
        warnings.warn('Setting update_collection to None will make u being updated every W execution. '
                      'This maybe undesirable. Please consider using a update collection instead.')
    # u is overwritten before backward runs -> the kernels read a snapshot
    u_read = K.clone(u.detach()) if update_collection != NO_OPS else u.detach()
    (W_bar,), batch = Fn.spectral_norm_batch([W], [u_read])
    _apply_update([u], batch, update_collection)
    if with_sigma:
        return W_bar, batch.sigma(0)
    return W_bar


def _trunc_normal(rng, size):
    """tf.truncated_normal_initializer() (sn.py:32)."""
    out = rng.normal(size=size)
    bad = abs(out) > 2
    while bad.any():
        out[bad] = rng.normal(size=int(bad.sum()))
        bad = abs(out) > 2
    return out.astype('float32')


def sn_pairs(store, prefix, with_names=False):
    """[(W, u)] for every spectrally normalised variable under `prefix`, by the reference's names:
    conv `X/filters/spectral_norm/u` <-> `X/Filters` (conv2d.py:142,170), linear
    `X/spectral_norm/u` <-> `X/W` (linear.py:140,162-164)."""
    pairs = []
    for name in store.names(prefix):
        if name.endswith('/filters/spectral_norm/u'):
            wname = name[:-len('/filters/spectral_norm/u')] + '/Filters'
        elif name.endswith('/spectral_norm/u'):
            wname = name[:-len('/spectral_norm/u')] + '/W'
        else:
            continue
        pairs.append((store.vars[wname], store.vars[name], wname) if with_names else (store.vars[wname], store.vars[name]))
    return pairs


def _flat_base(store, prefix, us):
    """The flat state buffer if `us` are exactly its consecutive views (ParamStore.flatten_state)."""
    f = store.flat.get(prefix + "#state")
    if f is None:
        return None
    buf, o = f["buf"], 0
    for u in us:
        if u.data_ptr() != buf.data_ptr() + 4 * o:
            return None
        o += u.numel()
    return buf if o == buf.numel() else None


@contextlib.contextmanager
def precomputed(store, prefix, update_collection=None, prepare=True, prep_kind=None, label_dense=None):
    """Normalise every SN weight under `prefix` in ONE batched launch group; inside the block
    `spectral_normed_weight(W, ...)` returns the precomputed W_bar for those W.  Also (prepare=True)
    builds the bf16 MFMA operand layouts of all W_bar in one launch and hands every W_bar a pre-zeroed
    slice of one flat gradient buffer, so the backward pass needs no per-weight memsets.
    `prep_kind(variable_name, W)` -> 0 | 1 | 2 | None chooses the operand layout per weight (kernels.prep_weights_batched).
    label_dense = (embedding table, name of a dense weight under `prefix`, bias | None): the per-label rows of that dense
    layer on the table come out of the same launches (`W_bar._label_T`, functional.concat_label).
    Round 3: normalisation, operand copies and label table are ONE launch pair (gank_sn_power_iter_fwd_prep)."""
    pairs = sn_pairs(store, prefix, with_names=True)
    if not pairs:
        yield None
        return
    Ws = [w for w, _, _ in pairs]
    us = [u for _, u, _ in pairs]
    flat = _flat_base(store, prefix, us)
    kinds = None
    if prepare:
        kinds = [prep_kind(nm, w) for w, _, nm in pairs] if prep_kind is not None else [0] * len(pairs)
    prep = (kinds, True) if prepare and len(pairs) <= 16 else None
    label = None
    if label_dense is not None and len(pairs) <= 16:
        tab, wname, bias = label_dense
        label = (tab, [nm for _, _, nm in pairs].index(wname), bias)
    # persistent workspaces of this network (kernels.SnState, registered by a trainer): a pass whose power iteration was already
    # run by the previous update's optimiser launch runs its second launch only
    state = getattr(store, "sn_state", {}).get(prefix)
    if state is not None and (len(pairs) > 16 or not state.matches(Ws, [u.detach() for u in us])):
        state = None
    if update_collection is None:
        # u <- u_final on every execution (sn.py:55-56): the kernels keep the u they read for the backward pass and
        # write u_final over u themselves -- no snapshot copy, no copy-back
        W_bars, batch = Fn.spectral_norm_batch(Ws, [u.detach() for u in us], snapshot=True, inplace=True, prep=prep, label=label, state=state)
    elif update_collection == NO_OPS and state is not None:
        W_bars, batch = Fn.spectral_norm_batch(Ws, [u.detach() for u in us], prep=prep, label=label, state=state)       # u is read, never written
    else:
        if update_collection != NO_OPS:
            if flat is not None:                      # one snapshot copy instead of one per weight
                snap, o, u_read = K.clone(flat), 0, []
                for u in us:
                    u_read.append(snap[o:o + u.numel()])
                    o += u.numel()
            else:
                u_read = [K.clone(u.detach()) for u in us]
        else:
            u_read = [u.detach() for u in us]
        W_bars, batch = Fn.spectral_norm_batch(Ws, u_read, prep=prep, label=label)
        _apply_update(us, batch, update_collection)
    if prepare:
        if prep is None:           # more weights than one launch pair takes: the separate preparation launch
            K.prep_weights_batched(list(W_bars), want_d=True, kinds=kinds)
        if any(w.requires_grad for w in Ws):
            need = sum(w.numel() for w in W_bars)
            gflat = _grad_scratch[0]
            _grad_scratch[0] = None                      # one use per zeroing
            if gflat is None or gflat.numel() < need:
                gflat = K.zeros_f32(need, W_bars[0].device) if W_bars[0].is_cuda else torch.zeros(need, dtype=torch.float32, device=W_bars[0].device)
            o = 0
            for wb in W_bars:
                wb._grad_buf = gflat[o:o + wb.numel()].view(wb.shape)
                o += wb.numel()
    _active.append({id(w): (wb, batch.sigma(i)) for i, (w, wb) in enumerate(zip(Ws, W_bars))})
    try:
        yield batch
    finally:
        _active.pop()
