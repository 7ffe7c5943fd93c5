"""Training step of the NetVLAD path (reference: train.build_graph, train.py:193-345; utils.py:170-213).

One process per GPU.  The reference splits a global batch over in-graph towers, SUMs the tower
gradients on the host, clips every variable to L2 norm 1 and applies Adam (train.py:266-336).  Here a
rank is a tower: forward/backward run locally, gradients live in ONE flat fp32 arena per rank so the
cross-tower SUM is an RCCL all-reduce on arena slices (no packing copies), launched as early as
backward produces them (the 554 MB ``hidden1_weights`` gradient -- 85 % of the payload -- sits next
to the loss, so its all-reduce overlaps the whole encoder / NetVLAD backward), and the per-variable
clip + TF-style Adam is one fused multi-tensor HIP kernel over the arena.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import FLAGS, layers, losses, ops, utils
from . import variables as vs

ARENA_ALIGN = 4096   # LPM_ARENA_ALIGN: every variable starts on a chunk boundary of the optimizer kernel

# The collective this rank entered last (name, arena slice, state): what a watchdog prints when a multi-rank run stops making
# progress (bench.py --gpus N), so that a hang names the exchange it hangs in.
LAST_COLLECTIVE = "none yet"


def _note(what: str):
    global LAST_COLLECTIVE
    LAST_COLLECTIVE = what


def learning_rate(base_learning_rate, global_step, batch_size, num_towers, decay_examples, decay):
    """tf.train.exponential_decay(staircase=True) over examples seen (train.py:244-249)."""
    p = math.floor(global_step * batch_size * num_towers / decay_examples)
    return base_learning_rate * decay ** p


class ParameterArena:
    """Flat fp32 arenas (param / grad / adam m / adam v) with every trainable variable a view into them."""

    def __init__(self, store: vs.VariableStore, first: Optional[List[str]] = None, gather: bool = False, bucket_of=None):
        """gather=False: every variable's .grad IS its slice of the gradient arena (autograd accumulates in place; one tiny
        add kernel per variable per step).  gather=True: .grad stays None during backward (autograd adopts each producer's
        tensor) and ``collect`` moves all of them into the arena with one multi-tensor copy."""
        self.gather = gather
        tv = store.trainable_variables()
        names = [n for n in (first or []) if n in tv] + [n for n in tv if n not in (first or [])]
        if bucket_of is not None:             # stable sort: variables of one all-reduce bucket are contiguous in the arena
            names = sorted(names, key=bucket_of)
        self.names = names
        offs, cur = [], 0
        for n in names:
            offs.append(cur)
            cur += (tv[n].numel() + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
        self.total = cur
        offs.append(cur)
        dev = tv[names[0]].device
        self.device = dev
        self.param = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.m = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.v = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.offsets_host = offs
        self.offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.views: Dict[str, torch.Tensor] = {}
        self.grad_views: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for n, o in zip(names, offs):
                t = tv[n]
                pv = self.param[o:o + t.numel()].view(t.shape)
                pv.copy_(t)
                t.data = pv                                   # the variable now lives in the arena
                if not gather:
                    t.grad = self.grad[o:o + t.numel()].view(t.shape)
                self.views[n] = t
                self.grad_views[n] = self.grad[o:o + t.numel()].view(t.shape)
        store.frozen = True
        self._scratch = None
        self.direct = []          # (name, offset, numel): gradients their producer writes straight into the arena
        self.l2: Dict[str, float] = {}   # gather mode: analytic L2 penalties of the current step (see gather_names)
        self._name_of = {id(t): n for n, t in self.views.items()}

    def mark_direct(self, name: str, on_ready=None):
        """The op producing this variable's gradient writes it into the arena itself (ops._Projection) and returns no
        gradient to autograd: no memset, no accumulate pass, no 554 MB copy.  ``on_ready`` runs right after the write."""
        t = self.views[name]
        a0, _ = self.segment(name)
        t._lpm_grad_view = self.grad[a0:a0 + t.numel()].view(t.shape)
        t._lpm_grad_written = False
        t._lpm_grad_ready = on_ready
        self.direct.append((name, a0, t.numel()))

    def gather_names(self, names):
        """Copy the gradients of ``names`` (gather mode) into their arena slices and release them.  Runs on the CURRENT stream:
        a gradient that was allocated on another stream (the audio branch's side stream) is recorded with the allocator for
        this one before it is released -- ``wait_stream`` orders the copy behind the producer, but only ``record_stream``
        keeps the block from being handed out again on its home stream while the copy is still queued here.
        ``self.l2`` (name -> coefficient, set by the trainer for the current step): the analytic gradient
        ``coefficient * w`` of an L2 weight penalty is added in the arena right behind the copy, i.e. BEFORE the slice can
        leave in an all-reduce -- every tower's gradient carries the penalty, as in the reference (train.py:296-303,321)."""
        dst, src, l2_dst, l2_src, l2_coef = [], [], [], [], []
        cur = torch.cuda.current_stream() if self.device.type == "cuda" else None
        with torch.no_grad():
            for name in names:
                t, gv = self.views[name], self.grad_views[name]
                g = t.grad
                if g is None:
                    gv.zero_()
                elif g.is_contiguous() and g.dtype == gv.dtype:
                    dst.append(gv)
                    src.append(g)
                else:
                    gv.copy_(g)
                if g is not None and cur is not None:
                    g.record_stream(cur)
                t.grad = None
                coef = self.l2.get(name)
                if coef:
                    l2_dst.append(gv)
                    l2_src.append(t)
                    l2_coef.append(coef)
            if dst:
                torch._foreach_copy_(dst, src)
            for coef in sorted(set(l2_coef)):
                sel = [i for i, c in enumerate(l2_coef) if c == coef]
                torch._foreach_add_([l2_dst[i] for i in sel], [l2_src[i].detach() for i in sel], alpha=coef)

    def collect(self, skip=()):
        """After backward: make the gradient arena complete (``skip``: names already gathered by a bucket hook).  Direct variables: a producer that was not reached leaves a
        zero gradient, anything autograd accumulated on the side is folded in.  gather mode: every other variable's
        gradient is copied into its arena slice (one multi-tensor copy for the contiguous ones)."""
        direct = set()
        for name, a0, n in self.direct:
            direct.add(name)
            t = self.views[name]
            if not t._lpm_grad_written:
                t._lpm_grad_view.zero_()
            if t.grad is not None:
                t._lpm_grad_view.add_(t.grad)
                t.grad = None
            coef = self.l2.get(name)
            if coef:                          # analytic L2 penalty of a directly written gradient: coefficient * w (see gather_names)
                t._lpm_grad_view.add_(t.detach(), alpha=coef)
        if not self.gather:
            return
        self.gather_names([n for n in self.names if n not in direct and n not in skip])

    def segment(self, name: str):
        i = self.names.index(name)
        return self.offsets_host[i], self.offsets_host[i + 1]

    def zero_grad(self):
        if self.gather:                       # nothing to clear: collect() overwrites every slice
            for name in self.names:
                self.views[name].grad = None
            for name, _, _ in self.direct:
                self.views[name]._lpm_grad_written = False
            return
        if not self.direct:
            self.grad.zero_()
            return
        cur = 0
        for name, a0, n in sorted(self.direct, key=lambda d: d[1]):
            if a0 > cur:
                self.grad[cur:a0].zero_()
            cur = a0 + n                      # (alignment padding after a direct segment stays zero forever)
            self.views[name].grad = None
            self.views[name]._lpm_grad_written = False
        if cur < self.total:
            self.grad[cur:].zero_()


class CollectiveTrace:
    """Per-collective overlap evidence for the first real multi-GPU run (bench.py --gpus N switches it on for the
    timed steps and prints the averages).  For every asynchronous collective of a step three events on the COMPUTE stream: at its
    launch, just before the step waits for it, and right after that wait.  window = launch -> before-wait: the compute the collective
    had to hide under; exposed = before-wait -> after-wait: how long the compute stream actually stood still for it.  A collective that
    is hidden has exposed ~ 0 whatever its own duration.  Off (the default) it costs nothing."""

    def __init__(self):
        self.on = False
        self.open = {}           # tag -> (MiB, launch mark)
        self.rows = []           # (tag, MiB, launch, before, after) of finished waits

    @staticmethod
    def _mark():
        """A time stamp on the compute stream: a HIP event on a GPU; the host clock on the CPU (the gloo tests run the same code path --
        there the 'compute stream' is the calling thread)."""
        if torch.cuda.is_available():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    @staticmethod
    def _ms(a, b):
        return a.elapsed_time(b) if not isinstance(a, float) else (b - a) * 1e3

    def launched(self, tag, nbytes):
        if self.on:
            self.open[tag] = (nbytes / 2.0 ** 20, self._mark())

    def wait(self, tag, work):
        if not self.on or tag not in self.open:
            work.wait()
            return
        mib, e0 = self.open.pop(tag)
        e1 = self._mark()
        work.wait()
        e2 = self._mark()
        self.rows.append((tag, mib, e0, e1, e2))

    def summary(self):
        """[{collective, MiB, calls, window_ms, exposed_ms}] averaged over the recorded steps (synchronises)."""
        if not self.rows:
            return []
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        acc = {}
        for tag, mib, e0, e1, e2 in self.rows:
            a = acc.setdefault(tag, [mib, 0, 0.0, 0.0])
            a[1] += 1
            a[2] += self._ms(e0, e1)
            a[3] += self._ms(e1, e2)
        self.rows = []
        return [{"collective": t, "MiB": round(a[0], 1), "calls": a[1], "window_ms": round(a[2] / a[1], 3), "exposed_ms": round(a[3] / a[1], 3)}
                for t, a in acc.items()]


TRACE = CollectiveTrace()


class GradientSynchronizer:
    """Cross-rank SUM of the gradient arena (utils.combine_gradients semantics: sum, not mean).
    Buckets are contiguous arena slices; ``early`` buckets are all-reduced from an autograd hook as soon
    as their last gradient is accumulated, the rest after backward.  Works with any torch.distributed
    backend (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, arena_grad: torch.Tensor, buckets: List[tuple], group=None):
        self.grad = arena_grad
        self.buckets = buckets
        self.group = group
        self.pending = []
        self.done = set()

    @property
    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def launch(self, i: int):
        if not self.active or i in self.done:
            return
        a, b = self.buckets[i]
        if b <= a:
            self.done.add(i)
            return
        _note(f"all_reduce(SUM) of gradient bucket {i} = arena[{a}:{b}] ({4 * (b - a) >> 20} MiB): launching")
        self.pending.append((i, dist.all_reduce(self.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        TRACE.launched(f"all_reduce bucket {i}", 4 * (b - a))
        _note(f"all_reduce(SUM) of gradient bucket {i} = arena[{a}:{b}] ({4 * (b - a) >> 20} MiB): launched, not yet waited for")
        self.done.add(i)

    def finish(self):
        for i in range(len(self.buckets)):
            self.launch(i)
        for i, w in self.pending:
            _note(f"all_reduce(SUM) of gradient bucket {i}: waiting for completion")
            TRACE.wait(f"all_reduce bucket {i}", w)
        if self.pending:
            _note(f"all_reduce(SUM) of gradient buckets {[i for i, _ in self.pending]}: complete")
        self.pending, self.done = [], set()


class ShardedVariableUpdate:
    """Route C of DESIGN.md section 6 for ONE large variable (hidden1_weights, 85 % of the parameters): ZeRO-1 over the towers.

    utils.combine_gradients SUMs the towers' gradients (utils.py:207-211), clip_gradient_norms clips the summed gradient of every
    variable to L2 norm ``clip`` (:181-188), Adam applies it (train.py:330-336).  For this variable the sum is a REDUCE-SCATTER of its
    slice of the gradient arena -- every rank receives the summed gradient of its 1/N shard, half the bytes of an all-reduce on the
    critical path --, the variable's norm is the one-double all-reduce of the shards' sums of squares, every rank runs clip + Adam on
    its shard only (1/N of the optimiser's HBM traffic), and the updated shards are ALL-GATHERed into every rank's parameter arena
    asynchronously: the next step's forward waits for them where it first reads the variable (VariableStore.pending), i.e. the
    gather rides under frame preparation, pooling and the encoders.  Every rank ends a step with bit-identical parameters by
    construction: each shard is computed once, by its owner.  Adam moments exist on the owner only (``gather_moments`` before a
    checkpoint).  Works on any torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, arena: ParameterArena, name: str, group=None, adam_fn=None):
        self.arena, self.name, self.group = arena, name, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        a0, a1 = arena.segment(name)
        if not self.supported(arena, name, self.world):
            raise ValueError(f"{name}: its arena segment of {a1 - a0} floats does not divide into {self.world} chunk-aligned shards")
        self.a0, self.a1 = a0, a1
        self.shard = (a1 - a0) // self.world
        self.lo = a0 + self.rank * self.shard
        self.hi = self.lo + self.shard
        self.gshard = torch.zeros(self.shard, dtype=torch.float32, device=arena.device)      # the summed gradient of the shard
        self.adam_fn = adam_fn                   # (p, g, m, v, lr, step) -> None, in place; None: ops.clip_adam_step
        self._rs = None
        self._ag = None
        self._ag_src = None
        self._scratch = None
        self._offsets = torch.tensor([0, self.shard], dtype=torch.int64, device=arena.device)
        self.launched = False
        self.last_norm = None                    # 0-dim tensor: the variable's gradient norm of the latest step (diagnostics, tests)
        self.keep_summed = False                 # tests: keep a copy of the shard's summed, un-clipped gradient (``summed_shard``)
        self.summed_shard = None
        # gloo has no reduce-scatter for device tensors (the shared-GPU debug mode): there the same sum arrives as an all-reduce of the
        # segment of which this rank keeps its shard -- decided from the backend, i.e. identically on every rank
        self._native_rs = not (dist.get_backend(group) == "gloo" and arena.device.type != "cpu")

    @staticmethod
    def supported(arena: ParameterArena, name: str, world: int) -> bool:
        """Shards must be whole optimiser chunks (ARENA_ALIGN floats) and equal: the collective's input is exactly the segment."""
        if name not in arena.views or world < 2:
            return False
        a0, a1 = arena.segment(name)
        return (a1 - a0) % (world * ARENA_ALIGN) == 0

    def launch(self):
        """The variable's gradient is complete in the arena: start the reduce-scatter (called by the gradient's producer through
        ParameterArena.mark_direct, or by the trainer after backward when no producer ran)."""
        if self.launched:
            return
        self.launched = True
        seg = self.arena.grad[self.a0:self.a1]
        _note(f"reduce_scatter(SUM) of {self.name}'s gradient = arena[{self.a0}:{self.a1}] ({4 * (self.a1 - self.a0) >> 20} MiB) into "
              f"{self.world} shards: launching")
        if self._native_rs:
            self._rs = dist.reduce_scatter_tensor(self.gshard, seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._rs = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        TRACE.launched("reduce_scatter hidden1_weights", 4 * (self.a1 - self.a0))
        _note(f"reduce_scatter(SUM) of {self.name}'s gradient: launched, not yet waited for")

    def step(self, clip: float, lr: float, global_step: int):
        """clip_by_norm + Adam on this rank's shard, then the parameter all-gather (left in flight: ``wait_parameters``)."""
        a = self.arena
        self.wait_parameters()                   # (a step without a forward in between: never in the trainer)
        if not self.launched:
            self.launch()
        _note(f"reduce_scatter(SUM) of {self.name}'s gradient: waiting for completion")
        TRACE.wait("reduce_scatter hidden1_weights", self._rs)
        self._rs = None
        self.launched = False
        if not self._native_rs:
            self.gshard.copy_(a.grad[self.lo:self.hi])
        g = self.gshard
        if self.keep_summed:
            self.summed_shard = g.clone()
        ss = torch.linalg.vector_norm(g, dtype=torch.float64).square().reshape(1)        # this shard's share of ||g||^2
        _note(f"all_reduce(SUM) of {self.name}'s shard norms (one double)")
        dist.all_reduce(ss, op=dist.ReduceOp.SUM, group=self.group)                       # identical on every rank
        norm = ss.sqrt()
        self.last_norm = norm[0]
        if clip and clip > 0:                                                             # utils.py:181-188: g * clip / max(norm, clip)
            g.mul_((clip / torch.clamp(norm, min=clip)).to(torch.float32)[0])
        p, m, v = a.param[self.lo:self.hi], a.m[self.lo:self.hi], a.v[self.lo:self.hi]
        if self.adam_fn is not None:
            self.adam_fn(p, g, m, v, lr, global_step)
        else:
            from . import ops
            # (clip_norm 0: the shard is clipped already -- the kernel's own norm would be the shard's, not the variable's)
            self._scratch = ops.clip_adam_step(p, g, m, v, self._offsets, 1, 0.0, lr, global_step, scratch=self._scratch)
        # the updated shard goes out from a private copy: the destination is the whole segment, this rank's shard included
        self._ag_src = p.clone()
        _note(f"all_gather of {self.name}'s updated parameter shards ({4 * self.shard >> 20} MiB each): launching")
        self._ag = dist.all_gather_into_tensor(a.param[self.a0:self.a1], self._ag_src, group=self.group, async_op=True)
        TRACE.launched("all_gather hidden1_weights parameters", 4 * (self.a1 - self.a0))
        _note(f"all_gather of {self.name}'s updated parameter shards: launched, waited for by the next read of the variable")

    def wait_parameters(self):
        if self._ag is not None:
            _note(f"all_gather of {self.name}'s updated parameter shards: waiting for completion")
            TRACE.wait("all_gather hidden1_weights parameters", self._ag)
            _note(f"all_gather of {self.name}'s updated parameter shards: complete")
            self._ag = self._ag_src = None

    def gather_moments(self):
        """A collective (every rank calls it): every rank receives the owners' Adam moments of the variable -- before a checkpoint
        that must hold them in full (Trainer.state_dict(sync=True))."""
        a = self.arena
        for arr, what in ((a.m, "Adam m"), (a.v, "Adam v")):
            src = arr[self.lo:self.hi].clone()
            _note(f"all_gather of {self.name}'s {what} shards (ShardedVariableUpdate.gather_moments)")
            dist.all_gather_into_tensor(arr[self.a0:self.a1], src, group=self.group)


def dp_bucket_of(name: str) -> int:
    """All-reduce buckets in the order backward completes them: 1 head (MoE, gating, hidden1_bn) -> 0 hidden1_weights ->
    2 encoders -> 3 NetVLAD pooling + input_bn.  (0 comes first in the arena: the optimiser kernel wants it chunk-aligned
    at the front and it is 85 % of the payload.)"""
    if name.endswith("hidden1_weights"):
        return 0
    if "_attention/" in name or "cluster_attention" in name:
        return 2
    if "_VLAD/" in name or "input_bn" in name:
        return 3
    return 1


class BucketGather:
    """Data parallelism in gather mode: the moment every variable of a bucket has its gradient (autograd post-accumulate
    hooks, counted per step), the bucket is copied into the gradient arena and its all-reduce is launched -- the head
    bucket rides under the whole backward, the encoder bucket under the NetVLAD backward; only the small pooling bucket is
    exposed.  Gradients of the audio branch are produced on the side stream: the gather runs on the main stream after it
    has been made to wait for every stream that was used."""

    def __init__(self, arena: ParameterArena, sync: "GradientSynchronizer", bucket_names: Dict[int, List[str]], early: List[int]):
        self.arena, self.sync = arena, sync
        self.bucket_names = bucket_names
        self.early = [b for b in early if bucket_names.get(b)]
        self.owner = {n: b for b in self.early for n in bucket_names[b]}
        self.seen: Dict[int, set] = {}
        self.gathered = set()
        self.main_stream = None
        for n, b in self.owner.items():
            arena.views[n].register_post_accumulate_grad_hook(lambda p, n=n, b=b: self._hook(n, b))

    def arm(self):
        self.seen = {b: set() for b in self.early}
        self.gathered = set()
        if self.arena.device.type == "cuda":
            self.main_stream = torch.cuda.current_stream()

    def _hook(self, name, b):
        seen = self.seen.get(b)
        if seen is None or b in self.gathered:
            return
        seen.add(name)
        if len(seen) < len(self.bucket_names[b]):
            return
        self.gathered.add(b)
        if self.main_stream is not None:
            from . import ops
            cur = torch.cuda.current_stream()          # the stream of the node whose gradient completed the bucket
            with torch.cuda.stream(self.main_stream):
                if cur != self.main_stream:
                    self.main_stream.wait_stream(cur)
                for side in ops._SIDE_STREAMS.values():
                    self.main_stream.wait_stream(side)
                self.arena.gather_names(self.bucket_names[b])
                self.sync.launch(b)
        else:
            self.arena.gather_names(self.bucket_names[b])
            self.sync.launch(b)

    def gathered_names(self):
        return {n for b in self.gathered for n in self.bucket_names[b]}


class Trainer:
    """Owns the variable store, the arenas and the optimiser state; ``step`` is one ``sess.run(train_op)``."""

    def __init__(self, model, vocab_size=3862, batch_size=None, base_learning_rate=None, learning_rate_decay=None,
                 learning_rate_decay_examples=None, regularization_penalty=None, clip_gradient_norm=None,
                 label_loss_fn=None, device="cuda", group=None, seed=0, model_kwargs=None):
        self.model = model
        self.vocab_size = vocab_size
        self.batch_size = batch_size or FLAGS.batch_size
        self.base_lr = FLAGS.base_learning_rate if base_learning_rate is None else base_learning_rate
        self.lr_decay = FLAGS.learning_rate_decay if learning_rate_decay is None else learning_rate_decay
        self.lr_decay_examples = (FLAGS.learning_rate_decay_examples if learning_rate_decay_examples is None
                                  else learning_rate_decay_examples)
        self.reg_penalty = FLAGS.regularization_penalty if regularization_penalty is None else regularization_penalty
        self.clip = FLAGS.clip_gradient_norm if clip_gradient_norm is None else clip_gradient_norm
        self.loss_fn = label_loss_fn or losses.CrossEntropyLoss()
        self.device = torch.device(device)
        self.group = group
        self.store = vs.VariableStore(device=self.device, seed=seed)
        self.store.analytic_l2 = self.device.type == "cuda"       # L2 weight penalties enter as gradients (see step)
        self._l2_regs = []
        self.model_kwargs = dict(model_kwargs or {})
        self.global_step = 0
        self.arena: Optional[ParameterArena] = None
        self.sync: Optional[GradientSynchronizer] = None
        self.bucket_gather = None
        self.factored = None
        self.w16 = None              # ops.ComputeCopy of hidden1_weights (netvlad_storage='bf16' with the factored update)
        self._update_stream, self._update_joined = None, True
        self._poisoned = None        # the exception of a step that failed behind hidden1_weights' early update (_check_not_poisoned)
        self.sharded: Optional[ShardedVariableUpdate] = None
        self.weight_pack = ops.WeightPack() if self.device.type == "cuda" else None
        # FLAGS.dense_arithmetic: the encoders' dense GEMMs on fp16 planes (three-term forward, two-term input gradients, one-term weight
        # gradients) with the delayed per-tensor scales this object measures
        self.operand_scales = None
        # NetVladV2 as well (round 5, measured): what its logits batch norm amplifies is a FORWARD error, and the fp16 forward keeps all
        # three terms on 11 + 11-bit planes -- more exact than split-bf16; with q / k / v on fp16 planes the untouched-initialisation case
        # went from 6.3e-3 to 4.7e-3 worst gradient, the prepared-weights and benched cases hold 3e-4 (LPM_V2_FP16=0: split-bf16, A/B)
        fp16_models = ("NetVladV1",) if os.environ.get("LPM_V2_FP16") == "0" else ("NetVladV1", "NetVladV2")
        if (self.device.type == "cuda" and FLAGS.dense_arithmetic == "fp16x2" and type(model).__name__ in fp16_models
                and os.environ.get("LPM_DENSE_ARITHMETIC", "fp16x2") == "fp16x2"):
            self.operand_scales = ops.OperandScales(self.device)
        if self.device.type == "cuda":
            ops.enable_library_gemm_selection()

    @property
    def num_towers(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    # ---------------------------------------------------------------------------------------------
    def _forward(self, model_input, num_frames, labels, **kw):
        with vs.use_store(self.store):
            with vs.variable_scope("tower"):
                fused = labels is not None and type(self.loss_fn) is losses.CrossEntropyLoss
                result = self.model.create_model(model_input, num_frames=num_frames, vocab_size=self.vocab_size,
                                                 labels=labels, fused_cross_entropy=fused, **{**self.model_kwargs, **kw})
            reg_losses = self.store.pop_regularization_losses()
            self._l2_regs = self.store.pop_l2_regularizers()
        return result, reg_losses

    def _normalize_input(self, raw, num_frames=None):
        if raw.dtype == torch.uint8:          # quantised reader output: dequantise + pad + normalise in one pass
            if raw.is_cuda:
                return ops.dequantize_l2_normalize(raw, num_frames)
            t = torch.arange(raw.shape[1], device=raw.device).view(1, -1, 1)
            x = torch.where(t < num_frames.view(-1, 1, 1), utils.Dequantize(raw.float()), torch.zeros((), device=raw.device))
            return layers.l2_normalize(x, 2)
        if raw.is_cuda and raw.shape[-1] % 4 == 0 and raw.shape[-1] <= 2048 and not raw.requires_grad:
            return ops.l2_normalize_rows(raw)
        return layers.l2_normalize(raw, 2)

    def build(self, model_input_raw, num_frames, labels):
        """Create every variable (a throw-away forward: moving statistics are restored afterwards),
        then move the trainable ones into the flat arenas and set up the gradient buckets."""
        if self.arena is not None:
            return
        with torch.no_grad():
            x = self._normalize_input(model_input_raw.to(self.device), num_frames.to(self.device))
            # undo the moving-average side effects of the dry run: statistics that existed before it (e.g. loaded after a
            # predict()) get their values back, the ones it created start at their initial values
            before = {n: v.clone() for n, v in self.store.vars.items() if not self.store.trainable[n]}
            self._forward(x, num_frames, labels)
            self._l2_regs = []
            for n, v in self.store.vars.items():
                if n in before:
                    v.copy_(before[n])
                elif n.endswith("/moving_mean"):
                    v.zero_()
                elif n.endswith("/moving_variance"):
                    v.fill_(1.0)
        gather = self.device.type == "cuda"
        self.arena = ParameterArena(self.store, first=["tower/hidden1_weights"], gather=gather, bucket_of=dp_bucket_of)
        bucket_names: Dict[int, List[str]] = {}
        for n in self.arena.names:
            bucket_names.setdefault(dp_bucket_of(n), []).append(n)
        ranges = []
        for b in range(4):                                # contiguous arena slice per bucket (empty buckets: empty slice)
            names = bucket_names.get(b, [])
            if names:
                ranges.append((self.arena.segment(names[0])[0], self.arena.segment(names[-1])[1]))
            else:
                ranges.append((ranges[-1][1], ranges[-1][1]) if ranges else (0, 0))
        self.sync = GradientSynchronizer(self.arena.grad, ranges, self.group)
        # hidden1_weights' gradient is complete right after the projection GEMM's backward: start its all-reduce
        # there and let it ride under the encoder / NetVLAD backward.
        early = (lambda: self.sync.launch(0)) if self.sync.active else None
        h1 = "tower/hidden1_weights"
        self.factored = None
        self.w16 = None              # ops.ComputeCopy of hidden1_weights (netvlad_storage='bf16' with the factored update)
        self.sharded = None
        # beyond hidden1_factored_max_towers (or from hidden1_sharded_min_towers on): route C, the sharded update
        shard_from = FLAGS.hidden1_sharded_min_towers or (FLAGS.hidden1_factored_max_towers + 1)
        want_sharded = (self.sync.active and FLAGS.hidden1_sharded_update and self.num_towers >= shard_from
                        and h1 in self.arena.views and self.arena.names[0] == h1
                        and ShardedVariableUpdate.supported(self.arena, h1, self.num_towers))
        want_factored = (self.device.type == "cuda" and FLAGS.hidden1_factored_update and not want_sharded
                         and self.num_towers <= FLAGS.hidden1_factored_max_towers
                         and h1 in self.arena.views and self.arena.names[0] == h1
                         and self.arena.views[h1].dim() == 2 and self.arena.views[h1].shape[1] % 32 == 0)
        if self.sync.active:
            # The route of hidden1_weights' gradient selects between different collectives, so it is decided ONCE, here, from
            # the per-rank batch size every rank was built with, and agreed across ranks: factored only if every rank wants it and
            # all ranks hold the same 16-multiple of clips; sharded only if every rank wants it.  (Single rank: a step that does not
            # fit falls back by itself.)
            b = int(model_input_raw.shape[0])
            flag = torch.tensor([1 if (want_factored and b % 16 == 0) else 0, b, -b, 1 if want_sharded else 0], dtype=torch.int64,
                                device=self.device)
            _note("all_reduce(MIN) of the hidden1 gradient-route agreement (Trainer.build)")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            lo, hi = int(flag[1]), -int(flag[2])
            want_factored = bool(flag[0]) and lo == hi
            want_sharded = bool(flag[3])
        if want_sharded:
            # the projection's backward writes the gradient into the arena and starts the reduce-scatter from its callback; the
            # projection's forward of the NEXT step waits for the parameter all-gather (VariableStore.pending)
            self.sharded = ShardedVariableUpdate(self.arena, h1, self.group)
            n1 = self.arena.offsets_host[1]
            self._tail_offsets = (self.arena.offsets[1:] - n1).contiguous()
            self._tail_scratch = None
            if self.device.type == "cuda":
                self.arena.mark_direct(h1, on_ready=self._sharded_ready)
            else:
                self.arena.views[h1].register_post_accumulate_grad_hook(lambda p: self._sharded_ready())
        elif want_factored:
            # hidden1_weights' gradient is consumed as the product it is (ops.FactoredGradient, lpm_factored_clip_adam): the towers
            # exchange its two skinny factors instead of all-reducing the gradient, which is never written
            self.factored = ops.FactoredGradient(on_put=self._factored_put, strict=self.sync.active)
            self.arena.views[h1]._lpm_factored = self.factored
            if FLAGS.netvlad_storage == "bf16" and FLAGS.hidden1_compute_copy and self.device.type == "cuda":
                # master fp32 + bf16 compute copy (+2 bytes per weight: 1.1 GB at cfg-5); the update pass keeps it current
                self.w16 = ops.ComputeCopy(self.arena.views[h1], also=(self.arena.param,))
                self.arena.views[h1]._lpm_w16 = self.w16
            n1 = self.arena.offsets_host[1]
            self._tail_offsets = (self.arena.offsets[1:] - n1).contiguous()
            self._factored_scratch = self._tail_scratch = None
            self._factored_work = []
        elif self.device.type == "cuda" and h1 in self.arena.views:
            self.arena.mark_direct("tower/hidden1_weights", on_ready=early)
        if self.device.type == "cuda" and not self.sync.active and FLAGS.direct_weight_gradients:
            # single GPU: the encoders' dense kernels receive their gradients straight from the split-K sums of their weight-gradient
            # GEMMs (ops._dw_x3) -- no fresh gradient tensor, no AccumulateGrad copy of a column view, no gather copy into the arena.
            # (Data parallel: the early buckets are counted from autograd hooks, which a direct write never fires -- unchanged there.)
            for n in self.arena.names:
                enc = n.endswith("/kernel") and "_attention/" in n
                moe = n in ("tower/gates/weights", "tower/experts/weights")          # ops.linear_direct (video_level_models.MoeModel)
                if (enc or moe) and self.arena.views[n].dim() == 2 and n not in {d[0] for d in self.arena.direct}:
                    self.arena.mark_direct(n)
        if (early is not None and self.sharded is None and self.factored is None and self.device.type != "cuda"
                and h1 in self.arena.views):
            # in-place arenas (CPU / gloo): bucket 0 leaves from the variable's post-accumulate hook.  (On the GPU the projection's
            # callback -- mark_direct's on_ready -- or the factor gather does it; a second launcher here would fire before the
            # accumulate of a weight used twice: ADVICE r3.)
            self.arena.views[h1].register_post_accumulate_grad_hook(lambda p: early())
        # the head and encoder buckets are gathered + all-reduced from hooks as backward completes them
        import os
        early_buckets = [int(b) for b in os.environ.get("LPM_DP_EARLY_BUCKETS", "1,2").split(",") if b != ""]
        self.bucket_gather = (BucketGather(self.arena, self.sync, bucket_names, early=early_buckets)
                              if (gather and self.sync.active and early_buckets) else None)
        if self.sync.active:
            # every rank must start from identical weights (the reference shares variables across towers)
            _note("broadcast of the parameter arena from rank 0 (Trainer.build)")
            dist.broadcast(self.arena.param, src=0, group=self.group)
            for n, v in self.store.vars.items():
                if not self.store.trainable[n]:
                    dist.broadcast(v, src=0, group=self.group)
            self.invalidate_compute_copies()

    def _check_not_poisoned(self):
        if self._poisoned is not None:
            raise RuntimeError("this Trainer is in an inconsistent state: a step failed AFTER hidden1_weights (and its Adam moments) had "
                               "been advanced inside backward (FLAGS.hidden1_early_update) while global_step and every other variable "
                               f"stayed behind -- restore() a checkpoint or build a new Trainer.  The step failed with: {self._poisoned!r}")

    def step(self, model_input_raw, num_frames, labels, **kw):
        """One optimiser step on this rank's shard of the global batch.  Returns loss / predictions."""
        self._check_not_poisoned()
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        dev = self.device
        model_input_raw = model_input_raw.to(dev)
        labels = labels.to(dev)
        num_frames = num_frames.to(dev)
        self.build(model_input_raw, num_frames, labels)
        self.arena.zero_grad()
        if self.bucket_gather is not None:
            self.bucket_gather.arm()
        model_input = self._normalize_input(model_input_raw, num_frames)                        # train.py:262-264
        if self.operand_scales is not None:
            self.operand_scales.begin_step()       # harvest the maxima measured so far, decide this step's operand format
            ops._ACTIVE_SCALES = self.operand_scales
        if self.weight_pack is not None:
            # every dense weight's operand forms for this step's forward AND backward in one launch (ops.WeightPack); a sharded
            # hidden1_weights is not among them (its all-gather may still be in flight: only the projection reads it)
            self.weight_pack.begin_step()
            ops._ACTIVE_PACK = self.weight_pack
        try:
            return self._step_body(model_input_raw, model_input, num_frames, labels, kw)
        except BaseException as e:
            # the early update makes a step non-atomic -- ~85 % of the parameters may already be at step t + 1.  A retried
            # step would apply Adam to them twice and a checkpoint written from an exception handler would be inconsistent: refuse both.
            early = getattr(self, "_early", None)
            if early is not None and early.get("done"):
                self._poisoned = e
            raise
        finally:
            ops._ACTIVE_SCALES = None
            if self.weight_pack is not None:
                ops._ACTIVE_PACK = None
                self.weight_pack.end_step()

    def _step_body(self, model_input_raw, model_input, num_frames, labels, kw):
        result, reg_losses = self._forward(model_input, num_frames, labels, **kw)
        predictions = result["predictions"]
        label_loss = result["loss"] if "loss" in result else self.loss_fn.calculate_loss(predictions, labels)  # :291-294
        reg_loss = result.get("regularization_loss", 0.0)
        if reg_losses:
            reg_loss = reg_loss + torch.stack(reg_losses).sum()                                 # :301-303
        final_loss = self.reg_penalty * reg_loss + label_loss                                   # :321
        # slim.l2_regularizer penalties recorded by the forward (store.analytic_l2, the GPU trainer): d/dw [penalty * scale *
        # sum(w^2)/2] = penalty * scale * w is added where the gradient enters the arena (ParameterArena.gather_names), which
        # for the early buckets happens inside backward, before their all-reduce leaves.
        self.arena.l2 = {}
        direct = {d[0] for d in self.arena.direct}
        for w, scale in self._l2_regs:
            name = self.arena._name_of.get(id(w))
            if name is None or not self.arena.gather or (name in direct and self.sync.active):
                raise RuntimeError("analytic L2 penalty on a variable the gather-mode arena does not gather")
            self.arena.l2[name] = self.arena.l2.get(name, 0.0) + self.reg_penalty * scale
        self._l2_regs = []
        # One tower on the GPU: the penalties' gradients are formed inside the clip + Adam passes from the parameter they read anyway
        # (lpm_multi_tensor_clip_adam_l2) instead of by add passes over the gradient arena (68 us per cfg-5 step for the MoE matrices).
        # Towers > 1 keep the add: it has to be in the arena before the all-reduce leaves.
        self._l2_fold = None
        fold_l2 = FLAGS.fold_l2_into_update
        if os.environ.get("LPM_L2_FOLD") in ("0", "1"):                # A/B
            fold_l2 = os.environ["LPM_L2_FOLD"] == "1"
        if (self.arena.l2 and fold_l2 and self.device.type == "cuda" and not self.sync.active and self.sharded is None
                and not (self.factored is not None and self.arena.names[0] in self.arena.l2)):
            key = tuple(sorted(self.arena.l2.items()))
            if getattr(self, "_l2_key", None) != key:
                vec = torch.zeros(len(self.arena.names), dtype=torch.float32)
                for n, c in self.arena.l2.items():
                    vec[self.arena.names.index(n)] = c
                self._l2_vec, self._l2_key = vec.to(self.device), key
            self._l2_fold = self._l2_vec
            self._l2_folded = dict(self.arena.l2)      # (Trainer.gradient puts the penalty back for whoever asks for the raw gradient)
            self.arena.l2 = {}               # (nothing left for ParameterArena.collect / gather_names to add)
        else:
            self._l2_folded = {}
        fg = self.factored
        if fg is not None:
            fg.clear()
            fg.armed = True
            fg.fold_dx = False
        # One tower: hidden1_weights is updated INSIDE backward, right behind the projection's input gradient -- its clip is per
        # variable (utils.py:181-188) and needs only the two factors the projection's backward has just handed over, and nothing reads
        # the old weight after dx.  The 0.58 ms update pass then fills the stretch in which the host is still enqueueing the ~80 small
        # launches of the audio encoder's backward and the main queue would run nothing (0.35-0.6 ms per cfg-2 step, tools/step_gaps.py).
        self._early = None
        if fg is not None and not self.sync.active and FLAGS.hidden1_early_update:
            lr0 = learning_rate(self.base_lr, self.global_step, model_input_raw.shape[0], self.num_towers, self.lr_decay_examples, self.lr_decay)
            self._early = {"lr": lr0, "step": self.global_step + 1, "done": False}
            fold = FLAGS.hidden1_fold_input_gradient
            if os.environ.get("LPM_FOLD_DX") in ("0", "1"):                # A/B
                fold = os.environ["LPM_FOLD_DX"] == "1"
            fg.fold_dx = bool(fold and self.w16 is not None)               # (offered; the projection's backward takes it if the shape fits)
        try:
            final_loss.backward()                                                               # :322-323
        finally:
            if fg is not None:
                fg.armed = False
        factored = fg is not None and fg.pending
        skip = set(self.bucket_gather.gathered_names()) if self.bucket_gather is not None else set()
        if factored:
            skip.add(self.arena.names[0])
        self.arena.collect(skip=skip)
        if self.sharded is not None:
            self._sharded_ready()             # (no-op when the projection's backward started the reduce-scatter already)
        self.sync.finish()                                                                      # utils.combine_gradients :330
        lr = learning_rate(self.base_lr, self.global_step, model_input_raw.shape[0], self.num_towers,
                           self.lr_decay_examples, self.lr_decay)                               # :244-249
        self.global_step += 1
        self._l2_lr = (lr, self.global_step)
        if self.sharded is not None:
            # route C: every other variable as usual, hidden1_weights on this rank's shard; its parameter all-gather stays in flight
            # until the next read of the variable (the projection of the next forward, a checkpoint, predict())
            a = self.arena
            n1 = a.offsets_host[1]
            if len(a.names) > 1:
                self._tail_scratch = ops.clip_adam_step(a.param[n1:], a.grad[n1:], a.m[n1:], a.v[n1:], self._tail_offsets,
                                                        len(a.names) - 1, self.clip, lr, self.global_step, scratch=self._tail_scratch)
            self.sharded.step(self.clip, lr, self.global_step)                                  # :332-336 for hidden1_weights
            self.store.pending[a.names[0]] = self.sharded.wait_parameters
        elif factored:
            self._factored_finish()
            a = self.arena
            n1, k = a.offsets_host[1], a.views[a.names[0]].numel()
            if len(a.names) > 1:
                self._tail_scratch = ops.clip_adam_step(a.param[n1:], a.grad[n1:], a.m[n1:], a.v[n1:], self._tail_offsets,
                                                        len(a.names) - 1, self.clip, lr, self.global_step, scratch=self._tail_scratch,
                                                        l2=self._l2_fold[1:] if self._l2_fold is not None else None)
            early = self._early
            if early is not None and early["done"]:
                assert early["lr"] == lr and early["step"] == self.global_step     # (the update ran inside backward with these)
            else:
                self._factored_scratch = fg.clip_adam(a.param[:k], a.m[:k], a.v[:k], self.clip, lr, self.global_step,
                                                      scratch=self._factored_scratch, param_bf16=self._w16_current())   # :332-336 for hidden1_weights
            self._early = None
        else:
            if self.w16 is not None:
                self.w16.invalidate()          # the generic update writes the master through raw pointers: the copy is rebuilt at its next use
            self.arena._scratch = ops.clip_adam_step(self.arena.param, self.arena.grad, self.arena.m, self.arena.v,
                                                     self.arena.offsets, len(self.arena.names), self.clip, lr,
                                                     self.global_step, scratch=self.arena._scratch, l2=self._l2_fold)  # :332-336
        self._join_update_stream()
        return {"loss": label_loss.detach(), "predictions": predictions.detach(), "learning_rate": lr,
                "global_step": self.global_step}

    def calibrate_operand_scales(self, model_input_raw, num_frames, labels, **kw):
        """The fp16 two-product format's scales measured NOW, synchronously, from one forward + backward on this batch in split-bf16
        (no optimiser step; moving statistics restored): the next ``step`` runs in fp16 with scales that fit it.  Without this call a
        run's first steps stay on split-bf16 until the asynchronous measurements arrive (two or three steps).  Single tower only (the
        backward's collectives are not entered here).  No-op for models / flags without operand scales."""
        sc = self.operand_scales
        if sc is None:
            return False
        if self.sync is not None and self.sync.active:
            raise RuntimeError("calibrate_operand_scales: single tower only (data-parallel runs calibrate over their first steps)")
        dev = self.device
        model_input_raw, labels, num_frames = model_input_raw.to(dev), labels.to(dev), num_frames.to(dev)
        self.build(model_input_raw, num_frames, labels)
        before = {n: v.clone() for n, v in self.store.vars.items() if not self.store.trainable[n]}
        self.arena.zero_grad()
        model_input = self._normalize_input(model_input_raw, num_frames)
        self._early = None                       # (no optimiser work inside this backward)
        was = sc.enabled
        sc.enabled = False                       # this pass runs in split-bf16 whatever is known so far
        sc.begin_step()
        ops._ACTIVE_SCALES = sc
        fg = self.factored
        try:
            result, reg_losses = self._forward(model_input, num_frames, labels, **kw)
            loss = result["loss"] if "loss" in result else self.loss_fn.calculate_loss(result["predictions"], labels)
            reg = result.get("regularization_loss", 0.0)
            if reg_losses:
                reg = reg + torch.stack(reg_losses).sum()
            if fg is not None:
                fg.clear()
                fg.armed = True
            (self.reg_penalty * reg + loss).backward()
        finally:
            ops._ACTIVE_SCALES = None
            sc.enabled = was
            self._l2_regs = []
            if fg is not None:
                fg.armed = False
                fg.clear()
        sc.calibrate_from_device()
        with torch.no_grad():
            for n, v in before.items():
                self.store.vars[n].copy_(v)
        for v in self.arena.views.values():      # direct-write marks of the calibration backward
            if hasattr(v, "_lpm_grad_written"):
                v._lpm_grad_written = False
        return True

    def gradient(self, name: str) -> torch.Tensor:
        """The raw (summed over towers, un-clipped) gradient of variable ``name`` that the latest ``step`` consumed: its slice of the
        gradient arena, or -- for hidden1_weights when the factored update ran, whose gradient is never written -- the product of
        the factors the optimiser used (tests, diagnostics).  On the sharded route the summed gradient of hidden1_weights exists only
        shard by shard on its owners (the arena slice holds this rank's LOCAL gradient after a native reduce-scatter): asking for it
        raises -- ``sharded.keep_summed`` / ``sharded.summed_shard`` give this rank's shard of the sum."""
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        t = self.arena.views[name]
        if self.sharded is not None and name == self.arena.names[0]:
            raise RuntimeError(f"gradient({name!r}): on the sharded route the summed gradient exists only as the owners' shards "
                               f"(set trainer.sharded.keep_summed = True and read trainer.sharded.summed_shard)")
        if self.factored is not None and self.factored.pending and name == self.arena.names[0]:
            return self.factored.materialise().view(t.shape)
        a0, _ = self.arena.segment(name)
        g = self.arena.grad[a0:a0 + t.numel()].view(t.shape)
        coef = getattr(self, "_l2_folded", {}).get(name)
        if coef:
            # the L2 penalty's share, coefficient * w, was added inside the clip + Adam passes (FLAGS.fold_l2_into_update) from the weight
            # BEFORE the update: w_old = w + lr_t m / (sqrt(v) + eps) with the moments the update left behind (exact to rounding)
            b1, b2, eps = 0.9, 0.999, 1e-8
            lr, step = self._l2_lr
            lr_t = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
            m = self.arena.m[a0:a0 + t.numel()].view(t.shape)
            v = self.arena.v[a0:a0 + t.numel()].view(t.shape)
            g = g + coef * (t.detach() + lr_t * m / (v.sqrt() + eps))
        return g

    def _update_stream_for(self, fg):
        """The HIP stream hidden1_weights' early update runs on (FLAGS.hidden1_update_stream): it waits for everything queued so far on the
        current stream -- the projection's input gradient, the factors' tiles -- and the step joins it before it returns
        (_join_update_stream): nothing on the main stream touches the variable, its moments or its compute copy in between."""
        if self.device.type != "cuda":
            return None
        if self._update_stream is None:
            self._update_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._update_stream.wait_stream(main)
        for t in (fg.xt, fg.dyt, fg.x, fg.dy, self._factored_scratch):
            if t is not None:
                t.record_stream(self._update_stream)
        self._update_joined = False
        return self._update_stream

    def _join_update_stream(self):
        if self._update_stream is not None and not self._update_joined:
            torch.cuda.current_stream().wait_stream(self._update_stream)
            self._update_joined = True

    def _w16_current(self):
        """hidden1_weights' bf16 compute copy for the update pass to rewrite (None: there is none, or it is stale and will be rebuilt)."""
        if self.w16 is None:
            return None
        return self.w16.current(self.arena.views[self.arena.names[0]])

    def _sharded_ready(self):
        """hidden1_weights' gradient is complete in the arena (called from the projection's backward, and again after backward)."""
        self.sharded.launch()
        self.sync.done.add(0)                 # bucket 0 is this variable: nothing left for the bucket all-reduce

    def _factored_put(self, fg):
        """Called inside backward when the projection has handed over its two gradient factors.  Data parallel: the towers' tile
        buffers are all-gathered (asynchronously, under the rest of backward) -- concatenated along their leading step axis they ARE
        the factors of the summed gradient (utils.combine_gradients) -- and bucket 0 has nothing left to all-reduce."""
        self._factored_work = []
        if not self.sync.active:
            early = getattr(self, "_early", None)
            if early is not None and not early["done"]:
                a = self.arena
                k = a.views[a.names[0]].numel()
                want = FLAGS.hidden1_update_stream
                want = (FLAGS.netvlad_storage == "bf16") if want == "auto" else bool(want)
                if os.environ.get("LPM_UPDATE_STREAM") in ("0", "1"):          # A/B
                    want = os.environ["LPM_UPDATE_STREAM"] == "1"
                w16 = self._w16_current()
                if fg.dx_out is not None:
                    # the projection's backward handed the factors over BEFORE forming its input gradient: dx rides in this pass, on the
                    # stream backward runs on -- or, should the pass not apply after all, nothing runs now: the projection then reads
                    # the old copy for dx first and the update follows backward (an update here would rewrite the copy under it)
                    if w16 is None or not fg.fold_supported():
                        return
                    self._factored_scratch = fg.clip_adam(a.param[:k], a.m[:k], a.v[:k], self.clip, early["lr"], early["step"],
                                                          scratch=self._factored_scratch, param_bf16=w16, dx=fg.dx_out)      # :332-336, early
                    fg.dx_done = True
                else:
                    us = self._update_stream_for(fg) if want else None
                    with (torch.cuda.stream(us) if us is not None else contextlib.nullcontext()):
                        self._factored_scratch = fg.clip_adam(a.param[:k], a.m[:k], a.v[:k], self.clip, early["lr"], early["step"],
                                                              scratch=self._factored_scratch, param_bf16=w16)   # :332-336, early
                early["done"] = True
                fg.early_done = True          # a second use of the weight in this backward must raise (ops._Projection.backward)
            return
        n = dist.get_world_size(self.group)
        xt_all = torch.empty(n * fg.xt.numel(), dtype=fg.xt.dtype, device=fg.xt.device)
        dyt_all = torch.empty(n * fg.dyt.numel(), dtype=fg.dyt.dtype, device=fg.dyt.device)
        for what, dst, src in (("descriptor", xt_all, fg.xt), ("output-gradient", dyt_all, fg.dyt)):
            _note(f"all_gather of the hidden projection's {what} tiles ({dst.numel() * 4 >> 20} MiB over {n} towers): launching")
            self._factored_work.append((what, dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)))
            TRACE.launched(f"all_gather {what} tiles", 4 * dst.numel())
        self._factored_all = (xt_all, dyt_all, n * fg.R, fg.xt, fg.dyt)         # (the local buffers stay alive until the wait)
        self.sync.done.add(0)

    def _factored_finish(self):
        if not self._factored_work:
            return
        for what, w in self._factored_work:
            _note(f"all_gather of the hidden projection's {what} tiles: waiting for completion")
            TRACE.wait(f"all_gather {what} tiles", w)
        _note("all_gather of the hidden projection's tiles: complete")
        fg = self.factored
        fg.xt, fg.dyt, fg.R = self._factored_all[:3]
        fg.x = fg.dy = None                  # the fp32 factors of the other towers are not here: the norm comes from the GEMM pass
        self._factored_work, self._factored_all = [], None

    # -- checkpoint / resume (train.py:501-515,593: Supervisor-saved variables + Adam slots + global_step) ------------------
    def state_dict(self, sync: bool = False) -> Dict[str, object]:
        """Everything a resumed run needs, keyed by the reference's TF variable names (``tower/video_VLAD/cluster_weights``,
        ``tower/video_attention/q/kernel``, ..., Adam slots as ``<name>/Adam`` and ``<name>/Adam_1``) plus ``global_step``.

        No communication by default: a chief-only ``if rank == 0: trainer.save(path)`` (the reference's Supervisor / is_chief
        pattern, train.py:501-515) must not enter a collective the other ranks never reach.  Data-parallel runs call
        ``prepare_checkpoint()`` on EVERY rank first (or pass ``sync=True`` on every rank): it averages the towers' moving statistics
        and, on the sharded route, gathers the owners' Adam moments of hidden1_weights -- the reference's chief holds ALL Adam slots.
        A chief-only save WITHOUT that on the sharded route holds hidden1_weights' moments for the chief's 1/N shard only; such a
        checkpoint is marked (``hidden1_adam_shard``) and ``load_state_dict`` refuses to resume from it."""
        self._check_not_poisoned()
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        if self.arena is None:
            raise RuntimeError("state_dict() needs a built trainer: run build() or one step first")
        if sync:
            self.sync_moving_statistics()
        if self.sharded is not None:
            self.sharded.wait_parameters()    # (this rank's own handle: no collective is entered)
            self.store.pending.pop(self.arena.names[0], None)
            if sync:
                self.sharded.gather_moments()
                self._moments_gathered_at = self.global_step
        out: Dict[str, object] = {n: v.detach().clone().cpu() for n, v in self.store.vars.items()}
        for n in self.arena.names:
            a0, _ = self.arena.segment(n)
            k = self.arena.views[n].numel()
            shape = self.arena.views[n].shape
            out[n + "/Adam"] = self.arena.m[a0:a0 + k].view(shape).clone().cpu()
            out[n + "/Adam_1"] = self.arena.v[a0:a0 + k].view(shape).clone().cpu()
        out["global_step"] = int(self.global_step)
        if self.sync is not None and self.sync.active and not sync:
            # a chief-only save: this rank's own moving statistics (the towers' mean needs sync_moving_statistics() on every rank
            # first) and, on the sharded route, Adam moments of hidden1_weights for this rank's shard only -- recorded, not hidden
            out["bn_statistics_synced"] = getattr(self, "_statistics_synced_at", None) == self.global_step
            if self.sharded is not None and getattr(self, "_moments_gathered_at", None) != self.global_step:
                out["hidden1_adam_shard"] = {"rank": self.sharded.rank, "towers": self.sharded.world,
                                             "floats": [self.sharded.lo - self.sharded.a0, self.sharded.hi - self.sharded.a0]}
        return out

    def prepare_checkpoint(self):
        """Data parallelism only (collectives: EVERY rank calls it, at the same step): what a chief-only ``save`` needs to hold the same
        state as the reference's chief (train.py:501-515) -- the towers' mean moving statistics and, on the sharded route, the owners'
        Adam moments of hidden1_weights in full on every rank.  ``if rank == 0: trainer.save(path)`` after it writes a complete,
        unmarked checkpoint."""
        self.sync_moving_statistics()
        self._statistics_synced_at = self.global_step
        if self.sharded is not None:
            self.wait_pending()
            self.sharded.gather_moments()
            self._moments_gathered_at = self.global_step

    def invalidate_compute_copies(self):
        """Call after ANY write into the parameters that does not go through torch's in-place ops on the variable or the arena -- a
        collective into ``arena.param`` (dist.broadcast / all_gather do not bump version counters), a ctypes kernel, a raw-pointer
        copy: hidden1_weights' bf16 compute copy (ops.ComputeCopy) is rebuilt at its next use."""
        if self.w16 is not None:
            self.w16.invalidate()

    def wait_pending(self):
        """Complete every asynchronous write into the variables (route C's parameter all-gather of hidden1_weights): call it before
        reading ``store.vars`` / ``arena.param`` directly between steps (``get_variable``, ``state_dict`` and ``load`` do it themselves)."""
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        self.store.drain_pending()

    def sync_moving_statistics(self):
        """Data parallelism only (a collective: every rank calls it).  Each rank is a tower with its own batch statistics
        and keeps its OWN moving averages during training (the reference lets every tower's update op write the shared
        variables in undefined order, train.py:309-316); before anything reads them across ranks -- a checkpoint, an
        evaluation -- they are re-synchronised to the mean over ranks, which is what one shared variable receiving every
        tower's ``decay``-weighted update converges to.  Call it on all ranks before a chief-only ``save`` and
        before a multi-rank ``predict`` (``state_dict(sync=True)`` does it too; the default ``state_dict()`` never communicates)."""
        if self.sync is None or not self.sync.active:
            return
        stats = [v for n, v in sorted(self.store.vars.items()) if not self.store.trainable[n]]
        if not stats:
            return
        with torch.no_grad():
            flat = torch.cat([v.reshape(-1) for v in stats])
            _note(f"all_reduce(SUM) of the batch-norm moving statistics ({flat.numel()} floats, Trainer.sync_moving_statistics)")
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            _note("all_reduce(SUM) of the batch-norm moving statistics: complete")
            flat /= self.num_towers
            off = 0
            for v in stats:
                v.copy_(flat[off:off + v.numel()].view(v.shape))
                off += v.numel()

    def load_state_dict(self, state: Dict[str, object]):
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        if self.arena is None:
            raise RuntimeError("load_state_dict() needs a built trainer: run build() first")
        if self.sharded is not None:
            self.sharded.wait_parameters()
            self.store.pending.pop(self.arena.names[0], None)
        shard = state.get("hidden1_adam_shard")
        if shard is not None:
            # a chief-only save on the sharded route without prepare_checkpoint() holds real Adam moments of hidden1_weights
            # for ONE rank's shard; resuming from it would run (N-1)/N of 85 % of the parameters with m = v = 0 under a bias correction
            # of ~1 (first updates ~ 3 lr sign(g)).  The reference's chief save holds all Adam slots (train.py:501-515).
            raise RuntimeError(
                f"checkpoint holds Adam moments of {self.arena.names[0]} for rank {shard.get('rank')}'s shard of {shard.get('towers')} only "
                f"(floats {shard.get('floats')}): it was saved by one rank of a sharded-update run without Trainer.prepare_checkpoint() "
                f"/ save(sync=True) on every rank first.  Re-save with the moments gathered, or delete 'hidden1_adam_shard' (and the "
                f"'/Adam', '/Adam_1' entries of that variable) from the state to restart its moments from zero knowingly.")
        with torch.no_grad():
            self.store.load({n: v for n, v in state.items() if n in self.store.vars}, strict=False)
            for v in self.store.vars.values():         # restored weights: the min |gamma| watch decides afresh, synchronously
                if hasattr(v, "_lpm_gamma_watch"):
                    del v._lpm_gamma_watch
            for n in self.arena.names:
                a0, _ = self.arena.segment(n)
                k = self.arena.views[n].numel()
                if n + "/Adam" in state:
                    self.arena.m[a0:a0 + k].copy_(torch.as_tensor(state[n + "/Adam"]).reshape(-1))
                if n + "/Adam_1" in state:
                    self.arena.v[a0:a0 + k].copy_(torch.as_tensor(state[n + "/Adam_1"]).reshape(-1))
        self.global_step = int(state.get("global_step", self.global_step))
        if self.w16 is not None:
            self.w16.invalidate()          # the master was rewritten (store.load copies through views: version counters see it; belt and braces)
        self._poisoned = None              # parameters, moments and global_step are one consistent state again

    def save(self, path: str, sync: bool = False):
        torch.save(self.state_dict(sync=sync), path)

    def restore(self, path: str):
        self.load_state_dict(torch.load(path, map_location="cpu"))

    @torch.no_grad()
    def predict(self, model_input_raw, num_frames, **kw):
        """eval.build_graph path: same forward with is_training=False (eval.py:143-150)."""
        self._join_update_stream()        # (hidden1_weights' update of an interrupted step may still be running on its stream)
        x = self._normalize_input(model_input_raw.to(self.device), num_frames.to(self.device))
        result, _ = self._forward(x, num_frames.to(self.device), None, is_training=False, **kw)
        return result["predictions"]
