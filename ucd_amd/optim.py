"""The optimiser of the train step: ``torch.optim.SGD`` with its step as one HIP launch.

The reference builds ``torch.optim.SGD(params, lr, momentum=0.9, nesterov=True)`` over three parameter groups with
weight decay (run.py:175-186) and calls ``optim.step()`` once per iteration (train.py:147).  :class:`SGD` keeps that class's
constructor, parameter groups, ``state`` / ``state_dict`` layout (``momentum_buffer`` per parameter) and hooks - schedulers
and checkpoints see a ``torch.optim.SGD`` - and replaces the step: ``ucd_sgd_step`` (csrc/sgd.hip) updates every tensor
in one launch and writes the bf16 working copies of the convolution weights (``ucd_amd.master``) in the same pass, so the
separate cast kernel after the step disappears as well.  The table of device pointers is built once and re-used while
the parameters, their gradients (views into the gradient buckets of ``ucd_amd.ddp``) and momentum buffers stay where
they are; a step only re-checks the addresses.

Not a CPU optimiser: parameters on the host raise.  Tensors the kernel cannot walk (gradient laid out differently from
its parameter, sparse or non-fp32 gradients) send that step through ``torch.optim.SGD.step`` on the device, with a warning.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import hip

MAX_GROUPS = 8          # UCD_SGD_MAX_GROUPS


class _SgdTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("w16", C.c_void_p), ("n", C.c_longlong),
                ("group", C.c_int), ("pad", C.c_int)]


class _SgdHyper(C.Structure):
    _fields_ = [("lr", C.c_double * MAX_GROUPS), ("momentum", C.c_double * MAX_GROUPS),
                ("weight_decay", C.c_double * MAX_GROUPS), ("nesterov", C.c_int * MAX_GROUPS)]


def _capturing():
    """Is the current stream being captured into a hipGraph?  (False on a host without a GPU, where asking raises.)"""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _same_layout(a, b):
    """Same element order in memory (the stride of a size-1 dimension is arbitrary: a 1x1 weight's differs between views)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def block_table(sizes, chunk):
    """[n_blocks, 2] int32 = {table entry, chunk index}: one workgroup per ``chunk`` elements of one tensor."""
    counts = (np.asarray(sizes, dtype=np.int64) + chunk - 1) // chunk
    owner_of = np.repeat(np.arange(counts.shape[0], dtype=np.int64), counts)
    first = np.repeat(np.cumsum(counts) - counts, counts)
    return np.ascontiguousarray(np.stack([owner_of, np.arange(owner_of.shape[0], dtype=np.int64) - first], axis=1).astype(np.int32))


class _Plan:
    __slots__ = ("signature", "moms", "table", "blocks", "n_blocks", "owners", "elements", "keep")


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, *, maximize=False,
                 foreach=None, differentiable=False, fused=None):
        if maximize or differentiable:
            raise NotImplementedError("ucd_amd.optim.SGD: maximize / differentiable are not part of the UCD train step")
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        self._plan = None
        self._warned = False
        self.refreshed_working_sets = ()      # Bf16Weights whose bf16 copies THIS step's kernel wrote (ucd_amd/master.py)
        # hyper-parameters in device memory (ucd_sgd_step_dev): what a captured hipGraph of the whole iteration replays while
        # the scheduler keeps changing the learning rate on the host (ucd_amd/train.py: Trainer._graph_step)
        self._hyper_dev = None

    # -- plan ----------------------------------------------------------------------------------------------------------
    def _signature(self):
        sig = []
        add = sig.append
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                add(p.data_ptr())
                add(0 if g is None else g.data_ptr())
        return sig

    def _moms_unchanged(self, plan):
        state = self.state
        for p, m in plan.moms:
            if state[p].get("momentum_buffer") is not m:
                return False
        return True

    def _build_plan(self, signature):
        """Pointer table + block table on the device, or None when some tensor cannot be walked by the kernel."""
        from .master import working_copy
        if len(self.param_groups) > MAX_GROUPS:
            return None
        rows, moms, owners, keep = [], [], [], []
        device = None
        for gi, group in enumerate(self.param_groups):
            if group.get("dampening", 0) != 0 or group.get("maximize", False):
                return None
            mu = group["momentum"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("ucd_amd.optim.SGD updates parameters on the GPU only (no CPU path)")
                if (p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse or g.device != p.device
                        or not _dense(p) or not _same_layout(g, p) or (device is not None and p.device != device)):
                    return None
                device = p.device
                m = None
                if mu != 0:
                    st = self.state[p]
                    m = st.get("momentum_buffer")
                    if m is None:
                        m = torch.zeros_like(p)               # mu*0 + g' = g': torch's first step (buffer = clone of the gradient)
                    elif m.dtype != torch.float32 or m.device != p.device or not _same_layout(m, p):
                        fresh = torch.empty_like(p)           # e.g. loaded from a reference checkpoint (contiguous NCHW)
                        fresh.copy_(m)
                        m = fresh
                    st["momentum_buffer"] = m
                    moms.append((p, m))
                w16, owner = working_copy(p)
                if w16 is not None:
                    if not _same_layout(w16, p) or w16.dtype != torch.bfloat16 or w16.device != p.device:
                        return None
                    if owner not in owners:
                        owners.append(owner)
                    keep.append(w16)
                rows.append((p.data_ptr(), g.data_ptr(), 0 if m is None else m.data_ptr(),
                             0 if w16 is None else w16.data_ptr(), p.numel(), gi))
        plan = _Plan()
        plan.signature, plan.moms, plan.owners, plan.keep = signature, moms, owners, keep
        plan.elements = sum(r[4] for r in rows)
        plan.n_blocks = 0
        plan.table = plan.blocks = None
        if not rows:
            return plan
        table = (_SgdTensor * len(rows))()
        for e, (pp, gp, mp, wp, n, gi) in zip(table, rows):
            e.p, e.g, e.m, e.w16, e.n, e.group = pp, gp, mp or None, wp or None, n, gi
        blocks = block_table([r[4] for r in rows], hip.load().ucd_sgd_chunk())
        plan.table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(device)
        plan.blocks = torch.from_numpy(blocks).to(device)
        plan.n_blocks = int(blocks.shape[0])
        return plan

    # -- hyper-parameters on the device ---------------------------------------------------------------------------------
    def _hyper(self):
        hyper = _SgdHyper()
        for gi, group in enumerate(self.param_groups):
            hyper.lr[gi] = float(group["lr"])
            hyper.momentum[gi] = float(group["momentum"])
            hyper.weight_decay[gi] = float(group["weight_decay"])
            hyper.nesterov[gi] = 1 if group["nesterov"] else 0
        return hyper

    def device_hyper(self, enable=True):
        """Switch the step to the device-resident hyper-parameter struct (a captured step graph needs it) or back."""
        if not enable:
            self._hyper_dev = None
            return
        if self._hyper_dev is None:
            dev = next(p for g in self.param_groups for p in g["params"]).device
            self._hyper_dev = torch.zeros(C.sizeof(_SgdHyper), dtype=torch.uint8, device=dev)
        self.push_hyper()

    def push_hyper(self):
        """Current lr / momentum / weight decay of every group -> the device struct, in stream order (one tiny launch)."""
        if self._hyper_dev is None:
            return
        hyper = self._hyper()
        with torch.cuda.device(self._hyper_dev.device):
            hip._check(hip.load().ucd_sgd_hyper_store(self._hyper_dev.data_ptr(), C.byref(hyper), hip.stream()), "ucd_sgd_hyper_store")

    def plan_is_current(self):
        """True when the next step() would launch the kernel without rebuilding its tables (no host-to-device copies)."""
        plan = self._plan
        return plan is not None and plan.signature == self._signature() and self._moms_unchanged(plan)

    # -- step ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.refreshed_working_sets = ()
        signature = self._signature()
        plan = self._plan
        if plan is None or plan.signature != signature or not self._moms_unchanged(plan):
            if _capturing():
                # building the tables copies host memory to the device: not capturable - the captured iteration must find the
                # tensors where the eager warm-up iterations left them (Trainer._graph_step checks plan_is_current() first)
                raise RuntimeError("ucd_amd.optim.SGD.step under graph capture: parameter / gradient / momentum tensors moved "
                                   "since the last eager step")
            plan = self._plan = self._build_plan(signature)
        if plan is None:
            if not self._warned:
                self._warned = True
                warnings.warn("ucd_amd.optim.SGD: a parameter / gradient layout the one-launch step cannot walk - "
                              "using torch.optim.SGD.step on the device for this optimiser")
            self._plan = None
            super().step()
            return loss
        if plan.n_blocks:
            capturing = _capturing()
            if capturing and self._hyper_dev is None:
                raise RuntimeError("ucd_amd.optim.SGD.step under graph capture needs device_hyper() first")
            with torch.cuda.device(plan.table.device):
                with hip._timed("ucd_sgd_step", 22.0 * plan.elements):
                    if self._hyper_dev is not None:
                        if not capturing:
                            self.push_hyper()       # a replayed graph gets its values from Trainer._graph_step before the launch
                        hip._check(hip.load().ucd_sgd_step_dev(plan.table.data_ptr(), plan.blocks.data_ptr(), plan.n_blocks,
                                                               self._hyper_dev.data_ptr(), hip.stream()), "ucd_sgd_step_dev")
                    else:
                        hyper = self._hyper()
                        hip._check(hip.load().ucd_sgd_step(plan.table.data_ptr(), plan.blocks.data_ptr(), plan.n_blocks,
                                                           C.byref(hyper), hip.stream()), "ucd_sgd_step")
            self.refreshed_working_sets = tuple(o for o in (ref() for ref in plan.owners) if o is not None)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plan = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plan = None
