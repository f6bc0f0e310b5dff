"""Name-keyed parameter store: the counterpart of TF1's variable scopes on the hot path.

The reference creates variables BY NAME on first use inside nested `tf.variable_scope`s and fetches
them again under `reuse=True` (common/ops/conv2d.py:59,142-144; SNGAN/gan_cifar_resnet.py:238,267);
callers never hold weight handles and select trainable sets by name substring (:507,:512).  The
store keeps the same names (`Generator/G.Block.1.Conv1/Filters`, `.../filters/spectral_norm/u`,
`Discriminator/D.Output/W`, ...) so a TF checkpoint name map is a dictionary lookup.

MI355X-first additions: `flatten(prefix)` re-homes all trainable variables of a network into ONE
fp32 buffer (plus one gradient buffer) so the optimiser is a single launch, gradient zeroing a single
memset and the data-parallel exchange a single large RCCL all-reduce over xGMI with no packing.
Weight gradients are accumulated by the kernels straight into `param.main_grad` views.
"""
import contextlib
from collections import OrderedDict

import numpy as np
import torch

_ALIGN = 4  # floats: every variable starts 16-byte aligned inside a flat buffer


class ParamStore:
    def __init__(self, device="cuda", seed=0):
        self.device = torch.device(device)
        self.vars = OrderedDict()       # full name -> tensor (fp32)
        self.trainable = OrderedDict()  # full name -> bool
        self.rng = np.random.RandomState(seed)
        self._scope = []
        self.flat = {}                  # prefix -> dict(params=, grads=, names=, offsets=)
        self.sn_state = {}              # prefix -> kernels.SnState: persistent spectral-norm workspaces a trainer registered (common/ops/sn.py)

    # ---- scopes -------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name, reuse=None):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return "/".join(self._scope + [name])

    # ---- variables ----------------------------------------------------------------------------
    def get_variable(self, name, shape=None, initializer=None, trainable=True):
        """tf.get_variable: create on first use (initializer = ndarray or callable(rng)->ndarray),
        fetch afterwards."""
        full = self.full_name(name)
        t = self.vars.get(full)
        if t is None:
            if initializer is None:
                raise KeyError(f"variable {full} does not exist and no initializer was given")
            val = initializer(self.rng) if callable(initializer) else initializer
            val = np.asarray(val, dtype=np.float32)
            if shape is not None and tuple(val.shape) != tuple(shape):
                raise ValueError(f"{full}: initializer shape {val.shape} != {tuple(shape)}")
            t = torch.from_numpy(val).to(self.device).contiguous()
            t.requires_grad_(bool(trainable))
            self.vars[full] = t
            self.trainable[full] = bool(trainable)
            for prefix in self.flat:
                if full.startswith(prefix.split("#")[0] + "/") and (trainable or prefix.endswith("#state")):
                    raise RuntimeError(f"{full} created after flatten({prefix!r}); build the graph once before flattening")
        return t

    def names(self, prefix=None, trainable=None):
        return [k for k in self.vars
                if (prefix is None or k.startswith(prefix + "/")) and (trainable is None or self.trainable[k] == trainable)]

    def params_with_name(self, substring):
        """`[var for var in tf.trainable_variables() if substring in var.name]` (gan_cifar_resnet.py:507,:512)"""
        return [v for k, v in self.vars.items() if substring in k and self.trainable[k]]

    def param_count(self, prefix):
        return sum(self.vars[k].numel() for k in self.names(prefix, trainable=True))

    # ---- state dict ---------------------------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.vars.items())

    def load_state_dict(self, state, strict=True):
        """Name+shape matching restore (cf. optimistic_restore, common/misc.py:275-307)."""
        for st in self.sn_state.values():
            st.valid = False                # weights / u change: a power iteration run ahead of time is stale
        with torch.no_grad():
            for k, v in state.items():
                if k not in self.vars:
                    if strict:
                        raise KeyError(k)
                    continue
                v = np.asarray(v, dtype=np.float32)
                if tuple(v.shape) != tuple(self.vars[k].shape):
                    if strict:
                        raise ValueError(f"{k}: shape {v.shape} != {tuple(self.vars[k].shape)}")
                    continue
                self.vars[k].copy_(torch.from_numpy(v).to(self.device))
                if hasattr(self.vars[k], "_prep"):
                    del self.vars[k]._prep          # cached MFMA operand copies are stale now

    # ---- flat buffers -------------------------------------------------------------------------
    def flatten(self, prefix, scratch_tail=False):
        """Move every trainable variable under `prefix` into one flat fp32 buffer; give each a
        `.main_grad` view into one flat gradient buffer.  Idempotent.  scratch_tail: the gradient buffer gets a
        second half (`scratch`) that zero_grads() clears in the same fill -- room for the gradients of derived
        weights (spectrally normalised copies), which would otherwise need their own fill per step."""
        if prefix in self.flat:
            return self.flat[prefix]
        names = self.names(prefix, trainable=True)
        offsets, total = {}, 0
        for k in names:
            offsets[k] = total
            total += (self.vars[k].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        params = torch.zeros(total, dtype=torch.float32, device=self.device)
        grads_all = torch.zeros(2 * total if scratch_tail else total, dtype=torch.float32, device=self.device)
        grads = grads_all[:total]
        with torch.no_grad():
            for k in names:
                v = self.vars[k]
                o, n = offsets[k], v.numel()
                params[o:o + n].copy_(v.reshape(-1))
                v.data = params[o:o + n].view(v.shape)
                v.main_grad = grads[o:o + n].view(v.shape)
        self.flat[prefix] = dict(params=params, grads=grads, names=names, offsets=offsets, grads_all=grads_all,
                                 scratch=grads_all[total:] if scratch_tail else None,
                                 m=torch.zeros_like(params), v=torch.zeros_like(params))
        for k in names:
            self.vars[k]._flat = self.flat[prefix]      # a kernel that accumulates OUTSIDE a backward pass marks the buffer dirty through this
        return self.flat[prefix]

    def flatten_state(self, prefix):
        """Pack the NON-trainable variables under `prefix` (the spectral-norm `u` vectors) into one flat
        buffer, in creation order, unpadded: snapshot and update of all of them are single copies."""
        key = prefix + "#state"
        if key in self.flat:
            return self.flat[key]
        names = self.names(prefix, trainable=False)
        total = sum(self.vars[k].numel() for k in names)
        buf = torch.zeros(total, dtype=torch.float32, device=self.device)
        o = 0
        with torch.no_grad():
            for k in names:
                v = self.vars[k]
                n = v.numel()
                buf[o:o + n].copy_(v.reshape(-1))
                v.data = buf[o:o + n].view(v.shape)
                o += n
        self.flat[key] = dict(buf=buf, names=names)
        return self.flat[key]

    def zero_grads(self, prefix):
        g = self.flat[prefix]["grads_all"]
        if g.is_cuda and g.dtype == torch.float32:
            from . import kernels as K
            K.zero_(g)                      # a kernel of the library (captured graphs hold kernel nodes only)
        else:
            g.zero_()


_default_store = None


def get_default_store():
    global _default_store
    if _default_store is None:
        _default_store = ParamStore("cuda" if torch.cuda.is_available() else "cpu")
    return _default_store


def set_default_store(store):
    global _default_store
    _default_store = store
    return store
