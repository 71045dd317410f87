"""Name-scoped variable store: the part of TensorFlow's variable_scope / get_variable
contract the reference's models rely on, over flat HBM arenas.

What callers of the reference's ops.py depend on (and this module provides):
  * variables are created BY NAME under the current scope and shared when the same name is
    requested again (dcgan.py:323-325, 413-414; cyclegan_module.py:9-14);
  * variable groups are picked by substring of the full name (dcgan.py:146-147,
    cyclegan.py:112-115);
  * the variable names themselves: w / biases (conv, deconv), Matrix / bias (linear),
    beta / gamma / moving_mean / moving_variance (batch norm).

Layout in HBM: after `pack()` every optimiser group owns four flat fp32 arenas (values,
gradients, Adam m, Adam v); a Variable's `.data` / `.grad` are views into them at 256-byte
aligned offsets.  One fused Adam launch and one all-reduce per group then cover every
variable of the group, and the wgrad kernels write gradients straight into the arena.
"""
import contextlib

import numpy as np
import torch

_ALIGN = 64  # floats: 256-byte slots keep every view 16-byte aligned for the vector loads


class Variable(object):
    def __init__(self, name, data, trainable):
        self.name = name
        self.data = data            # torch tensor on the device (a view into an arena after pack())
        self.grad = None            # same shape, view into the group's gradient arena
        self.trainable = trainable
        self.wants_grad = False     # set per update by the optimiser that owns the variable
        self.grad_fresh = True      # first gradient write of an update overwrites, later ones accumulate
        self.group = None
        self.early = False          # gradient complete at the backward pass's boundary (Group.split_early)

    @property
    def shape(self):
        return tuple(self.data.shape)

    def grad_target(self):
        """(tensor to write the gradient into, accumulate flag) for the current update."""
        if self.grad is None:
            self.grad = torch.empty_like(self.data)
        acc = 0 if self.grad_fresh else 1
        self.grad_fresh = False
        return self.grad, acc

    def __repr__(self):
        return "Variable(%s, %s)" % (self.name, self.shape)


class Group(object):
    """A var_list with its flat arenas (what tf.train.AdamOptimizer.minimize(var_list=...) owns)."""

    def __init__(self, name, variables, device):
        self.name = name
        self.variables = list(variables)
        offs, total = [], 0
        for v in self.variables:
            offs.append(total)
            total += (v.data.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.values = torch.zeros(total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=device)
        self.m = torch.zeros(total, dtype=torch.float32, device=device)
        self.v = torch.zeros(total, dtype=torch.float32, device=device)
        for var, off in zip(self.variables, offs):
            n = var.data.numel()
            view = self.values[off:off + n].view(var.data.shape)
            view.copy_(var.data)
            var.data = view
            var.grad = self.grads[off:off + n].view(var.data.shape)
            var.group = self
        self.offsets = offs
        self.early_offset = None

    def split_early(self, first_name):
        """Data parallel overlap: the variables from `first_name` to the end of the arena (the layers
        nearest the loss, whose gradients the backward pass finishes first) form the EARLY part, which
        is exchanged while the backward pass of the remaining layers still runs.  Must be called
        before the first taped forward pass."""
        names = [v.name for v in self.variables]
        if first_name not in names:
            raise KeyError("group %s has no variable %s" % (self.name, first_name))
        k = names.index(first_name)
        for i, v in enumerate(self.variables):
            v.early = i >= k
        self.early_offset = self.offsets[k]
        return self

    def early_grads(self):
        return self.grads[self.early_offset:]

    def late_grads(self):
        return self.grads[:self.early_offset]

    def finish_early(self):
        """At the boundary of a staged backward pass: early variables no gradient reached are
        zero-filled now, before their part of the arena is exchanged."""
        from . import _lib
        for v in self.variables:
            if v.early and v.grad_fresh:
                _lib.check(_lib.lib.gan_fill(_lib.ptr(v.grad), v.grad.numel(), 0.0, _lib.stream()))
                v.grad_fresh = False

    def broadcast(self, process_group=None, src=0):
        """Data parallel start-up: every rank takes rank `src`'s values (one collective per group)."""
        import torch.distributed as dist
        if self.values.is_cuda and dist.get_backend(process_group) == "gloo":
            host = self.values.cpu()            # gloo moves host memory (see AdamOptimizer.all_reduce)
            dist.broadcast(host, src=src, group=process_group)
            self.values.copy_(host)
        else:
            dist.broadcast(self.values, src=src, group=process_group)

    def begin_update(self):
        for v in self.variables:
            v.wants_grad = True
            v.grad_fresh = True

    def end_update(self):
        """Variables no gradient reached this update get a zero gradient (TF leaves them unchanged
        only if the gradient is None for the whole run; here the arena slot must not stay stale)."""
        from . import _lib
        for v in self.variables:
            if v.grad_fresh:
                _lib.check(_lib.lib.gan_fill(_lib.ptr(v.grad), v.grad.numel(), 0.0, _lib.stream()))
            v.wants_grad = False


class VariableStore(object):
    def __init__(self, device=None, seed=0):
        self.device = torch.device(device if device is not None else 'cuda')
        self.vars = {}              # creation order preserved (tf.trainable_variables() order)
        self._scope = []
        self._reuse = []
        self.rng = np.random.default_rng(seed)
        self.groups = {}

    # -- scoping ---------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name, reuse=None):
        self._scope.append(name)
        self._reuse.append(bool(reuse) or (self._reuse[-1] if self._reuse else False))
        try:
            yield self
        finally:
            self._scope.pop()
            self._reuse.pop()

    def reuse_variables(self):
        """scope.reuse_variables(): from here to the end of the innermost scope, missing names raise."""
        if self._reuse:
            self._reuse[-1] = True

    def full_name(self, name):
        return '/'.join(self._scope + [name])

    def get_variable(self, name, shape, initializer, trainable=True):
        """Create-or-share.  Inside a `reuse=True` scope a missing name raises ValueError as in
        TensorFlow; otherwise a missing name is created and an existing one is shared (the models
        run define-by-run, so the second step re-enters the scopes that created the variables)."""
        full = self.full_name(name)
        shape = tuple(int(s) for s in shape)
        var = self.vars.get(full)
        if var is None:
            if self._reuse and self._reuse[-1]:
                raise ValueError("Variable %s does not exist, or was not created with get_variable()" % full)
            host = np.asarray(initializer(shape, self.rng), dtype=np.float32).reshape(shape)
            var = Variable(full, torch.from_numpy(host).to(self.device), trainable)
            self.vars[full] = var
        elif var.shape != shape:
            raise ValueError("Trying to share variable %s, but specified shape %s and found shape %s"
                             % (full, shape, var.shape))
        return var

    # -- selection / packing ---------------------------------------------------------------
    def trainable_variables(self):
        return [v for v in self.vars.values() if v.trainable]

    def select(self, substr):
        """[var for var in tf.trainable_variables() if substr in var.name]"""
        return [v for v in self.trainable_variables() if substr in v.name]

    def pack(self, name, variables):
        if name in self.groups:
            raise ValueError("group %s already packed" % name)
        for v in variables:
            if v.group is not None:
                raise ValueError("variable %s already belongs to group %s" % (v.name, v.group.name))
        grp = Group(name, variables, self.device)
        self.groups[name] = grp
        return grp

    # -- host round trips (checkpoints, tests) -----------------------------------------------
    def state_dict(self):
        return {n: v.data.detach().cpu().numpy().copy() for n, v in self.vars.items()}

    def load_state_dict(self, state, strict=True):
        for n, arr in state.items():
            if n not in self.vars:
                if strict:
                    raise KeyError("no variable named %s" % n)
                continue
            v = self.vars[n]
            a = np.asarray(arr, dtype=np.float32)
            if tuple(a.shape) != v.shape:
                raise ValueError("variable %s: shape %s given, %s stored" % (n, a.shape, v.shape))
            v.data.copy_(torch.from_numpy(a).to(self.device))


# ---------------------------------------------------------------------------- initialisers
def truncated_normal_initializer(stddev):
    """tf.truncated_normal_initializer: values beyond 2 stddev are redrawn (ops.py:67)."""
    def init(shape, rng):
        out = rng.standard_normal(shape)
        bad = np.abs(out) > 2.0
        while bad.any():
            out[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(out) > 2.0
        return out * stddev
    return init


def random_normal_initializer(stddev):
    """tf.random_normal_initializer (ops.py:82, 109)."""
    return lambda shape, rng: rng.standard_normal(shape) * stddev


def constant_initializer(value):
    return lambda shape, rng: np.full(shape, value)


# ---------------------------------------------------------------------------- default store
_default = None


def default_store():
    global _default
    if _default is None:
        _default = VariableStore()
    return _default


def set_default_store(store):
    global _default
    _default = store
    return store


def variable_scope(name, reuse=None):
    return default_store().variable_scope(name, reuse)


@contextlib.contextmanager
def use_store(store):
    """Make `store` the default for the duration of a block (one model = one store)."""
    global _default
    old = _default
    _default = store
    try:
        yield store
    finally:
        _default = old
