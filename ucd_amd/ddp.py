"""Data-parallel gradient averaging over RCCL/xGMI, overlapped with the backward pass.

Replaces ``apex.parallel.DistributedDataParallel(model, delay_allreduce=True)`` (run.py:199,204,372),
which flattens every gradient after the whole backward and issues ONE blocking 232 MB all-reduce.
Here (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL):

* gradients live in flat, pre-allocated bucket buffers (``param.grad`` is a view), filled in reverse
  registration order - the order the backward produces them;
* the moment the last gradient of a bucket has been accumulated (post-accumulate-grad hook), the bucket
  is all-reduced (average) on a side HIP stream, so communication runs under the rest of the backward;
* ``finish()`` makes the compute stream wait for the outstanding buckets right before the optimiser;
* bucket size is sized for xGMI, not NVSwitch: the 8 GPUs of a node are fully connected by
  point-to-point links (7 x ~153 GB/s), a ring is bound by one link, so few large messages (default
  32 MB) amortise the per-collective latency better than many small ones; ``wire_dtype=torch.bfloat16``
  halves the bytes on the wire (gradients are averaged in bf16, then widened back).

The wrapper keeps the ``module.``-prefixed ``state_dict`` keys that the reference's checkpoints have
(run.py:37,217,222 save and load the DDP-wrapped model).

Drop-in use needs no extra calls: the reference's loop ``optim.zero_grad(); loss.backward(); optim.step()``
(train.py:104,137-138,149) works unchanged - the first gradient hook of a backward queues an end-of-backward callback
(``Variable._execution_engine.queue_callback``, the mechanism of torch's own DDP reducer) that completes and waits for the
buckets, and the wrapper's ``forward`` re-attaches the bucket views that ``optim.zero_grad(set_to_none=True)`` dropped.
``zero_grad()`` / ``finish_grad_sync()`` remain as the explicit (and idempotent) forms ``ucd_amd.train.Trainer`` uses.
Gradient accumulation over several backward passes per optimiser step is not supported (it raises).
"""
from __future__ import annotations

import weakref

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import switches as _switches


def _view_like(buf, p):
    """View of the flat slice ``buf`` with the shape AND strides of ``p`` (channels-last 4-D weights keep their
    format, so fused optimisers see matching layouts and autograd accumulates without re-striding)."""
    if p.dim() == 4:
        co, ci, kh, kw = p.shape
        v = buf.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        if v.stride() == p.stride():
            return v
    return buf.view_as(p)


class _Bucket:
    __slots__ = ("flat", "params", "pending", "work", "wire", "event", "fed", "done")


class GradReducer:
    def __init__(self, params, bucket_mb=32.0, wire_dtype=None, group=None, overlap=True, shadow_of=None, direct_modules=()):
        # shadow_of: {fp32 parameter: bf16 working copy that autograd differentiates} (ucd_amd/master.py); such a
        # parameter's slot in the fp32 bucket is filled from the bf16 gradient when the bucket completes
        self.shadow_of = shadow_of or {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # UCD_FORCE_COLLECTIVES=1 (bench.py --force_dist): a one-rank process group still launches every bucket's all-reduce - the
        # multi-rank code path, RCCL calls included, on a box with one GPU
        self.collective = self.world > 1 or (dist.is_initialized() and _switches.get("UCD_FORCE_COLLECTIVES") in ("1", "ddp"))
        # gradient buckets get a communicator of their own: they are launched from autograd hooks on a side
        # stream while InPlaceABNSync issues its statistics collectives on the compute stream - two independent
        # orderings that must not share one RCCL communicator
        if group is None and self.collective and dist.get_backend() == "nccl":
            group = dist.new_group(backend="nccl")
        self.group = group
        self.use_avg = self.collective and dist.get_backend(group) == "nccl"   # RCCL reduces with AVG; gloo sums
        self.wire_dtype = wire_dtype
        params = [p for p in params if p.requires_grad]
        # ABN layers whose backward kernels write [d bias | d weight] straight into gradient storage (ucd_amd/abn.py,
        # csrc/abn_node.cpp): their parameters get ONE flat bucket of their own, laid out bias-then-weight per layer, that
        # no autograd hook feeds - it is simply reduced with the stragglers in finish() (a few hundred KB)
        self.direct_flat = None
        direct = [m for m in direct_modules if getattr(m, "weight", None) is not None and m.weight.requires_grad
                  and m.bias is not None and m.bias.requires_grad and m.weight.is_cuda]
        if direct:
            total = sum(2 * m.weight.numel() for m in direct)
            self.direct_flat = torch.zeros(total, dtype=torch.float32, device=direct[0].weight.device)
            off, taken = 0, set()
            for m in direct:
                C = m.weight.numel()
                m.bias.grad = self.direct_flat[off:off + C]
                m.weight.grad = self.direct_flat[off + C:off + 2 * C]
                m._direct_grads = (self.direct_flat[off:off + 2 * C], self.direct_flat.data_ptr() + 4 * off)
                taken.add(m.weight); taken.add(m.bias)
                off += 2 * C
            params = [p for p in params if p not in taken]
            self._direct_modules = direct
        self.params = params
        self.device = params[0].device if params else torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        self.overlap = overlap and self.on_gpu and self.collective
        # fp32 buckets on an RCCL process group go through a library-owned communicator of their own (ucd_amd/comm.py:
        # ucd_comm_all_reduce_sum on the reducer's stream, then 1 / world) instead of c10d's ProcessGroupNCCL: the same call
        # path as the SyncBN exchanges, and one that a hipGraph capture of the whole step takes (c10d's asynchronous work
        # objects on a side stream crash hipStreamEndCapture on this stack - measured with bench.py --force_dist ddp)
        self._late_dst, self._late_src = [], []                 # world 1: bf16 -> fp32 gradient copies deferred to finish()
        self.direct = None
        # UCD_DDP_DIRECT: "auto" (default) takes the library-owned communicator where it has run - the one-rank forced-collectives
        # mode and a multi-rank run that captures its step (UCD_STEP_GRAPH=1: c10d's work objects cannot be captured) - and leaves a
        # plain multi-rank eager run on c10d's AVG all-reduce until a 2+ rank lockstep run of the direct path has passed;
        # "1" / "0" force it
        dd = _switches.get("UCD_DDP_DIRECT", "auto")
        want_direct = dd == "1" or (dd == "auto" and (self.world == 1 or _switches.get("UCD_STEP_GRAPH", "auto") == "1"))
        if (self.collective and self.on_gpu and wire_dtype in (None, torch.float32) and dist.get_backend(group) == "nccl"
                and _switches.get("UCD_DIRECT_RCCL", "1") != "0" and want_direct):
            from .comm import direct_comm
            self.direct = direct_comm(self.group, ipc=False, purpose="grad")
        self.stream = torch.cuda.Stream(self.device) if self.overlap else None
        self.buckets = []
        self._bucket_of = {}
        cap = int(bucket_mb * (1 << 20))
        cur, cur_bytes = [], 0
        for p in reversed(params):                       # backward order ~ reverse registration order
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > cap:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._make_bucket(cur)
        self._cb_queued = False
        self._finished = False          # this step's buckets are already reduced (finish() is idempotent)
        # a finished backward whose gradients were neither consumed by an optimiser step nor cleared: the next backward
        # would be gradient accumulation, which the bf16 working-copy hand-over and the kernel-written ABN gradients
        # (both overwrite) cannot express
        self._dirty = False
        ref = weakref.ref(self)

        def _after_step(optimizer, args, kwargs):
            me = ref()
            if me is not None:
                me._dirty = False
        self._step_hook = register_optimizer_step_post_hook(_after_step)
        self._events = {}
        self._hooks = []
        for p in params:
            holder = self.shadow_of.get(p)
            if holder is None:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            else:
                self._hooks.append(holder.register_post_accumulate_grad_hook(lambda t, p=p: self._on_grad(p)))
        self._inflight = []

    def _make_bucket(self, plist):
        b = _Bucket()
        total = sum(p.numel() for p in plist)
        b.flat = torch.zeros(total, dtype=plist[0].dtype, device=self.device)
        b.params, b.pending, b.work, b.wire, b.event = plist, len(plist), None, None, None
        b.fed, b.done = [], False
        off = 0
        for p in plist:
            p.grad = _view_like(b.flat[off:off + p.numel()], p)  # gradients accumulate straight into the bucket
            if p in self.shadow_of:
                b.fed.append((p.grad, self.shadow_of[p]))
            off += p.numel()
            self._bucket_of[p] = b
        self.buckets.append(b)

    # -- per step ------------------------------------------------------------------------------
    def zero_grad(self):
        """One memset per bucket instead of one per tensor; keeps the grad views alive."""
        self._finished = False
        self._dirty = False
        self.drop_pending_copies()
        if self.direct_flat is not None:
            self.direct_flat.zero_()
            off = 0
            for m in self._direct_modules:                      # re-attach views dropped by set_to_none
                C = m.weight.numel()
                if m.bias.grad is None:
                    m.bias.grad = self.direct_flat[off:off + C]
                if m.weight.grad is None:
                    m.weight.grad = self.direct_flat[off + C:off + 2 * C]
                off += 2 * C
        for b in self.buckets:
            # a bucket whose every slot is overwritten by the widening copy of its parameter's bf16 gradient needs no fill (round 6:
            # 232 MB of memsets per step at the bench size); a parameter that receives NO gradient in a step gets its slot zeroed
            # in _complete instead
            if len(b.fed) != len(b.params):
                b.flat.zero_()
            b.pending = len(b.params)
            b.done = False
            for _, holder in b.fed:
                holder.grad = None                              # autograd then adopts the incoming tensor
            off = 0                                             # re-attach views dropped by set_to_none
            for p in b.params:
                if p.grad is None:
                    p.grad = _view_like(b.flat[off:off + p.numel()], p)
                off += p.numel()

    def drop_pending_copies(self):
        """Forget widening copies queued by a backward that never reached finish() (an exception, a whole-step capture that failed
        mid-backward): their sources are tensors of a pass that did not complete - flushed together with the next step's copies they
        would land in the same bucket views in one multi-tensor launch, in undefined order."""
        self._late_dst, self._late_src = [], []
        if self._wgrad_mode():
            from . import hip
            hip.wgrad_drop()                                    # ... nor the slab sum its last weight gradient left pending

    def prepare_step(self):
        """Called by the wrapper's forward: if the caller cleared the gradients with ``optim.zero_grad()`` (set_to_none:
        the bucket views are gone) re-attach zeroed views, so that autograd accumulates into the buckets again and the ABN
        kernels can write their parameter gradients in place."""
        probe = self.params[0] if self.params else None
        dropped = probe is not None and probe.grad is None and self.shadow_of.get(probe) is None
        if not dropped and self.direct_flat is not None:
            dropped = self._direct_modules[0].bias.grad is None
        if not self._cb_queued:
            self.drop_pending_copies()                          # a new step begins: nothing queued by an aborted backward survives
        self._wgrad_defer(True)                                 # (drops a slab sum left pending by a backward that never finished)
        if dropped:
            self.zero_grad()

    def _wgrad_defer(self, on):
        """Deferred slab sums of the weight gradients (csrc/wgrad.hip, round 6): on for the backward pass of a step of this wrapper -
        the nodes' weight gradients are first READ by the bucket copies, and ``_wgrad_flush`` runs in front of those - off again in
        ``finish`` (a bare ``backward()`` outside the wrapper sums at once).  ``UCD_WGRAD_DEFER=0`` keeps the separate launches."""
        mode = self._wgrad_mode()
        if not mode:
            return
        from . import hip
        if on:
            hip.wgrad_drop()                                    # of a backward that never finished
        # bit 1: the nodes' weight gradients leave the compute stream for the library's side stream until the next flush (include/
        # ucd_hip.h) - nothing waits for them before the bucket copies, and off the chain of input-gradient products they cost the
        # small-batch step (3 - 6 images per GPU) a fifth of its dependent launches less
        hip.wgrad_defer(mode if on else 0)

    def _wgrad_mode(self):
        """ucd_conv_wgrad_defer's mode for this wrapper's backward passes: bit 0 deferred slab sums, bit 1 the side stream."""
        if not self.on_gpu:
            return 0
        return ((1 if _switches.get("UCD_WGRAD_DEFER", "1") != "0" else 0)
                | (2 if _switches.get("UCD_WGRAD_STREAM", "1") != "0" else 0))

    def _wgrad_flush(self):
        if self._wgrad_mode():
            from . import hip
            hip.wgrad_flush()

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if b.done or b.pending <= 0 or (self._dirty and not self._cb_queued):
            raise RuntimeError("ucd_amd.ddp: a second backward before optim.step() / zero_grad() - gradient accumulation over "
                               "several backward passes per optimiser step is not supported")
        if not self._cb_queued:
            # end-of-backward callback: finish() without any call from the training loop (apex DDP's allreduce hook)
            self._cb_queued = True
            self._finished = False
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        b.pending -= 1
        if b.pending == 0:
            self._complete(b)

    def _end_of_backward(self):
        self._cb_queued = False
        self.finish()

    def _complete(self, b):
        """All gradients of the bucket exist: widen the bf16 ones into their fp32 slots (one multi-tensor copy),
        then start the all-reduce."""
        if b.done:
            return
        b.done = True
        # gradients that autograd re-created outside the bucket (the caller dropped the views with set_to_none and no
        # forward of the wrapper re-attached them): move them in
        off = 0
        for p in b.params:
            n = p.numel()
            g = p.grad
            if g is not None and self.shadow_of.get(p) is None:
                view = _view_like(b.flat[off:off + n], p)
                if g.data_ptr() != view.data_ptr():
                    view.copy_(g)
                    p.grad = view
            elif g is None and self.shadow_of.get(p) is not None and self.shadow_of[p].grad is not None:
                p.grad = _view_like(b.flat[off:off + n], p)
            off += n
        if b.fed:
            dst, src = [], []
            for view32, holder in b.fed:
                if holder.grad is not None:
                    dst.append(view32)
                    src.append(holder.grad)
                elif len(b.fed) == len(b.params):
                    view32.zero_()                               # no gradient this step: the slot was not cleared by zero_grad
            if src and (self.collective or _switches.get("UCD_DDP_LATE_COPY", "1") == "0"):
                self._wgrad_flush()                              # the last weight gradient's slab sum may still be pending
                torch._foreach_copy_(dst, src)                   # before the bucket's reduction starts
            elif src:
                # a single process reduces nothing: the widening copies of ALL buckets go out as one multi-tensor launch at the end of
                # the backward (finish) instead of one per bucket (nine launches of ~25 us, mostly fixed cost, at the bench size)
                self._late_dst += dst
                self._late_src += src
            for _, holder in b.fed:
                holder.grad = None
        if self.collective:
            self._wgrad_flush()                                  # (no-op after the one above) a bucket's reduction never starts forked
            self._launch(b)

    def _event(self, key):
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        return ev

    def _launch(self, b):
        if self.overlap:
            ev = self._event(id(b))                              # one event per bucket, reused every step
            ev.record(torch.cuda.current_stream(self.device))    # bucket complete on the compute stream
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._reduce(b)
        else:
            self._reduce(b)
        self._inflight.append(b)

    def _direct_sum(self, buf):
        """In-place average of an fp32 buffer over the ranks on the CURRENT stream (library-owned communicator)."""
        from . import hip
        hip._check(hip.load().ucd_comm_all_reduce_sum(self.direct.handle, hip.ptr(buf), buf.numel(), hip.stream()),
                   "ucd_comm_all_reduce_sum")
        if self.world > 1:
            buf.mul_(1.0 / self.world)

    def _reduce(self, b):
        buf = b.flat
        if self.direct is not None and buf.dtype == torch.float32:
            self._direct_sum(buf)
            b.work = None
            return
        if self.wire_dtype is not None and self.wire_dtype != buf.dtype:
            b.wire = buf.to(self.wire_dtype)
            buf = b.wire
        if self.use_avg:
            b.work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:                                                   # gloo (tests): no AVG
            b.work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Block the compute stream (not the host) until every bucket of this step is averaged.  Runs from the
        end-of-backward callback; calling it again before the next backward is a no-op."""
        if self._finished:
            return
        self._finished = True
        self._dirty = True
        self._wgrad_flush()                                     # every weight gradient is final from here on
        self._wgrad_defer(False)
        for b in self.buckets:                                  # parameters that received no gradient
            if not b.done:
                self._complete(b)
        m0 = self._direct_modules[0] if self.direct_flat is not None else None
        if m0 is not None and (m0.bias.grad is None or m0.bias.grad.data_ptr() != self.direct_flat.data_ptr()):
            off = 0                                             # ABN gradients autograd produced outside the flat buffer
            for m in self._direct_modules:
                C = m.weight.numel()
                for p, o in ((m.bias, off), (m.weight, off + C)):
                    g = p.grad
                    if g is None:
                        # cleared between the forward (which handed the kernels this slice's address) and the backward
                        # (which wrote into it): hand the view back
                        p.grad = self.direct_flat[o:o + C]
                    elif g.data_ptr() != self.direct_flat.data_ptr() + 4 * o:
                        self.direct_flat[o:o + C].copy_(g)
                        p.grad = self.direct_flat[o:o + C]
                off += 2 * C
        if not self.collective:
            if self._late_src:
                torch._foreach_copy_(self._late_dst, self._late_src)
                self._late_dst, self._late_src = [], []
            self._reset_step()
            return
        if self.direct_flat is not None:                        # kernel-written ABN parameter gradients: one small reduce
            if self.overlap:
                ev = self._event("direct")
                ev.record(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.stream):
                    self.stream.wait_event(ev)
                    self._reduce_flat(self.direct_flat)
            else:
                self._reduce_flat(self.direct_flat)
        for b in self._inflight:
            if b.work is None:                                  # direct path: stream-ordered, nothing to wait for on the host
                continue
            if self.overlap:
                with torch.cuda.stream(self.stream):
                    b.work.wait()
                    if b.wire is not None:
                        b.flat.copy_(b.wire)
                    if not self.use_avg:
                        b.flat.div_(self.world)
            else:
                b.work.wait()
                if b.wire is not None:
                    b.flat.copy_(b.wire)
                if not self.use_avg:
                    b.flat.div_(self.world)
            b.work = b.wire = None
        if self.overlap:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self._inflight = []
        self._reset_step()

    def enable_direct(self):
        """Move the fp32 gradient buckets to the library-owned RCCL communicator (``UCD_DDP_DIRECT`` decided against it at
        construction: a plain multi-rank run keeps c10d's AVG all-reduce).  A collective: every rank calls it.  Returns True when
        the buckets now run on the direct path - what a captured multi-rank step needs (c10d's asynchronous work objects crash
        hipStreamEndCapture on this stack)."""
        if self.direct is not None:
            return True
        if not (self.collective and self.on_gpu and self.wire_dtype in (None, torch.float32) and dist.get_backend(self.group) == "nccl"
                and _switches.get("UCD_DIRECT_RCCL", "1") != "0" and _switches.get("UCD_DDP_DIRECT", "auto") != "0"):
            return False
        from .comm import direct_comm
        self.direct = direct_comm(self.group, ipc=False, purpose="grad")
        return self.direct is not None

    def _reset_step(self):
        """Per-step bucket state back to 'nothing arrived' (also done by zero_grad): a loop that never calls the reducer's
        zero_grad (the reference's optim.zero_grad()) must find fresh counters at its next backward."""
        for b in self.buckets:
            b.pending = len(b.params)
            b.done = False

    def _reduce_flat(self, flat):
        if self.direct is not None and flat.dtype == torch.float32:
            self._direct_sum(flat)
            return
        if self.use_avg:
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._step_hook.remove()


class DistributedDataParallel(nn.Module):
    """``DistributedDataParallel(model, delay_allreduce=True)`` call shape of apex (run.py:204).
    Parameters and buffers are broadcast from rank 0 at construction (apex does the same).
    The reference's loop works unchanged (``optim.zero_grad(); loss.backward(); optim.step()``): gradient averaging is
    finished by an end-of-backward callback.  ``zero_grad()`` (one memset per bucket) and ``finish_grad_sync()`` (idempotent)
    are the explicit forms."""

    def __init__(self, module, delay_allreduce=True, bucket_mb=32.0, wire_dtype=None, group=None, overlap=True,
                 bf16_weights=False):
        super().__init__()
        self.module = module
        self.bf16_weights = None
        self.delay_allreduce = delay_allreduce        # accepted for call compatibility; overlap decides
        self.reducer = None
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, src=0, group=group)
        params = [p for p in module.parameters() if p.requires_grad]
        shadow_of = None
        if bf16_weights and params and params[0].is_cuda:
            from .master import Bf16Weights
            self.bf16_weights = Bf16Weights(module)
            shadow_of = self.bf16_weights.shadow_of
        direct = [m for m in module.modules() if getattr(m, "ucd_fused_abn", False)] if (params and params[0].is_cuda) else []
        if params:
            self.reducer = GradReducer(params, bucket_mb, wire_dtype, group, overlap, shadow_of, direct)

    def forward(self, *args, **kwargs):
        if self.bf16_weights is not None:
            self.bf16_weights.refresh_if_stale()            # after an optimiser step / checkpoint load
        if self.reducer is not None and self.training and torch.is_grad_enabled():
            self.reducer.prepare_step()
        return self.module(*args, **kwargs)

    def zero_grad(self, set_to_none=False):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            super().zero_grad(set_to_none)

    def finish_grad_sync(self):
        if self.reducer is not None:
            self.reducer.finish()
