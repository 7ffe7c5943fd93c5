"""A small variable store standing in for TF1's graph collections + variable scopes.

The reference creates variables lazily by name inside ``tf.variable_scope`` blocks
(``tf.get_variable`` / ``slim`` layers, e.g. frame_level_models.py:2273-2277,2775-2808) and reuses them
across towers (train.py:276).  ``VariableStore`` reproduces that: the first ``get_variable`` under a
scope creates a leaf tensor on the current device, later calls return the same tensor.  Names follow
SURVEY.md App. A.9 so checkpoints / oracle weights can be exchanged by name.
"""
from __future__ import annotations

import contextlib
import math
from typing import Callable, Dict, List, Optional

import torch


class VariableStore:
    def __init__(self, device=None, seed: int = 0):
        self.device = torch.device(device) if device is not None else None
        self.vars: Dict[str, torch.Tensor] = {}
        self.trainable: Dict[str, bool] = {}
        self._scope: List[str] = []
        self._reg_losses: List[torch.Tensor] = []
        self._l2_regs = []                  # (variable, scale) pairs when analytic_l2 is set
        self.analytic_l2 = False
        self._gen_seed = seed
        self._gen: Optional[torch.Generator] = None
        self.frozen = False
        # tf.summary.histogram stand-in (frame_level_models.py:2780,2799; train.py:260,285): None = summaries off (nothing is
        # kept alive); a dict collects the named intermediate tensors of the next forward (detached)
        self.summaries: Optional[Dict[str, torch.Tensor]] = None
        # name -> callable run (once) by the next ``get_variable`` of that name, before the tensor is handed out: the data-parallel
        # trainer parks the wait for a parameter all-gather here (train.ShardedVariableUpdate), so that the collective rides under
        # the part of the next forward that does not read the variable
        self.pending: Dict[str, Callable] = {}

    # -- scopes ---------------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name: str):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name: str) -> str:
        return "/".join(self._scope + [name])

    # -- variables ------------------------------------------------------------------------------
    def _generator(self, device):
        if self._gen is None or self._gen.device != device:
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(self._gen_seed)
        return self._gen

    def get_variable(self, name: str, shape, initializer: Callable, trainable: bool = True, device=None) -> torch.Tensor:
        full = self.full_name(name)
        if full in self.vars:
            if self.pending:
                cb = self.pending.pop(full, None)
                if cb is not None:
                    cb()
            v = self.vars[full]
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"variable {full}: shape {tuple(v.shape)} != requested {tuple(shape)}")
            return v
        if self.frozen:
            raise RuntimeError(f"variable store is frozen (arena built); cannot create {full}")
        dev = torch.device(device) if device is not None else (self.device or torch.device("cpu"))
        with torch.no_grad():
            t = initializer(tuple(shape), dev, self._generator(dev)).to(torch.float32)
        t.requires_grad_(trainable)
        self.vars[full] = t
        self.trainable[full] = trainable
        return t

    def trainable_variables(self) -> Dict[str, torch.Tensor]:
        return {n: v for n, v in self.vars.items() if self.trainable[n]}

    def drain_pending(self):
        """Run every parked completion callback (``pending``: an asynchronous write into a variable that its next reader must wait for --
        route C's parameter all-gather).  ``get_variable`` does it per name; whole-store readers and writers do it here."""
        while self.pending:
            _, cb = self.pending.popitem()
            cb()

    def load(self, values: Dict[str, torch.Tensor], strict: bool = False):
        """Copy values in by name (e.g. the oracle's weight dict)."""
        self.drain_pending()
        with torch.no_grad():
            for n, v in values.items():
                if n in self.vars:
                    self.vars[n].copy_(v.reshape(self.vars[n].shape).to(self.vars[n].dtype))
                elif strict:
                    raise KeyError(n)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        self.drain_pending()
        return {n: v.detach().clone() for n, v in self.vars.items()}

    # -- regularisation collection (tf.losses.get_regularization_losses, train.py:301-303) ------
    def add_regularization_loss(self, t: torch.Tensor):
        self._reg_losses.append(t)

    def add_l2_regularizer(self, var: torch.Tensor, scale: float):
        """slim.l2_regularizer(scale)(var) = scale * sum(var^2) / 2.  With ``analytic_l2`` set (the GPU trainer) only
        (var, scale) is recorded and the trainer adds the gradient scale * var itself after backward -- the loss value is
        not needed there -- instead of building ~10 small kernels of graph per weight; otherwise the loss tensor is
        collected like any other regularisation loss."""
        if getattr(self, "analytic_l2", False):
            self._l2_regs.append((var, float(scale)))
        else:
            self._reg_losses.append(scale * 0.5 * (var * var).sum())

    def pop_l2_regularizers(self):
        out, self._l2_regs = self._l2_regs, []
        return out

    def pop_regularization_losses(self) -> List[torch.Tensor]:
        out, self._reg_losses = self._reg_losses, []
        return out


# -- initialisers (TF1 names) --------------------------------------------------------------------
def random_normal_initializer(stddev: float):
    return lambda shape, dev, gen: torch.randn(shape, device=dev, generator=gen) * stddev


def zeros_initializer():
    return lambda shape, dev, gen: torch.zeros(shape, device=dev)


def ones_initializer():
    return lambda shape, dev, gen: torch.ones(shape, device=dev)


def glorot_uniform_initializer():
    """tf.layers.dense default kernel init / slim xavier_initializer (uniform)."""
    def init(shape, dev, gen):
        fan_in, fan_out = shape[0], shape[-1]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, device=dev, generator=gen) * 2 - 1) * lim
    return init


# -- default store (TF's default graph) -----------------------------------------------------------
_default = VariableStore()


def default_store() -> VariableStore:
    return _default


@contextlib.contextmanager
def use_store(store: VariableStore):
    global _default
    prev, _default = _default, store
    try:
        yield store
    finally:
        _default = prev


def variable_scope(name: str):
    return _default.variable_scope(name)


def get_variable(name, shape, initializer, trainable=True, device=None):
    return _default.get_variable(name, shape, initializer, trainable, device)


def peek_variable(name: str):
    """The variable ``name`` under the current scope if it EXISTS already, else None -- never creates one (creation order is the
    reference's: a layer that wants to know the kernel of the layer behind it must not create that kernel early)."""
    return _default.vars.get(_default.full_name(name))


def summary(name: str, tensor: torch.Tensor):
    """Record an intermediate tensor of the forward under ``name`` when the current store collects summaries."""
    if _default.summaries is not None:
        _default.summaries[name] = tensor.detach()
