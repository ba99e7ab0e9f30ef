"""bf16 working copies of the convolution weights ("master weights" layout for --opt_level O1).

The reference runs apex AMP O1 (run.py:199-200): every convolution call casts its fp32 weight to the compute type,
and the backward casts the weight gradient back - two small kernels and two autograd nodes per layer and step,
~210 launches that matter once the step is launch-bound.  Here the fp32 weights the optimiser updates (and that
``state_dict`` saves) are re-pointed into ONE flat fp32 buffer, and a flat bf16 buffer of identical layout holds the
working copies the convolutions read; after an optimiser step the whole set is refreshed by a single cast kernel
(an optimiser-step hook marks them stale; other in-place writes are seen through the version counters).  The bf16 values are exactly the ones the
per-call cast would have produced (round-to-nearest-even of the fp32 master), so the forward is bit-identical.

The gradients of the working copies arrive in bf16; ``ucd_amd.ddp.GradReducer`` widens a whole bucket of them into
the fp32 bucket (= ``master.grad``) with one multi-tensor copy when the bucket is complete.
"""
from __future__ import annotations

import weakref

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook
from torch.utils.weak import WeakTensorKeyDictionary

from .blocks import Conv2d
from .ddp import _view_like


# fp32 master parameter -> (bf16 working copy, weakref to the Bf16Weights that owns it): how the one-launch optimiser step
# (ucd_amd/optim.py) finds the copy it writes together with the master
_WORKING = WeakTensorKeyDictionary()       # keyed by identity (a plain WeakKeyDictionary would compare tensors with ==)


def working_copy(param):
    """(bf16 working copy, weakref to its owner) of a trainable master weight, or (None, None)."""
    hit = _WORKING.get(param)
    if hit is None or hit[1]() is None:
        return None, None
    return hit


class Bf16Weights:
    def __init__(self, model, trainable=True):
        net = model.module if hasattr(model, "module") else model
        self.convs = [m for m in net.modules() if isinstance(m, Conv2d) and m.weight.is_cuda
                      and m.weight.dtype == torch.float32 and (m.weight.requires_grad or not trainable)]
        self.trainable = trainable
        self.shadow_of = {}
        if not self.convs:
            self.flat32 = self.flat16 = None
            return
        dev = self.convs[0].weight.device
        total = sum(m.weight.numel() for m in self.convs)
        self.flat32 = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat16 = torch.empty(total, dtype=torch.bfloat16, device=dev)
        off = 0
        with torch.no_grad():
            for m in self.convs:
                w, n = m.weight, m.weight.numel()
                v32 = _view_like(self.flat32[off:off + n], w)
                v32.copy_(w)
                w.data = v32                                  # the Parameter object (optimiser key) is unchanged
                s = _view_like(self.flat16[off:off + n], w)
                if trainable:
                    s.requires_grad_(True)
                m._w16 = s
                self.shadow_of[w] = s
                if trainable:
                    _WORKING[w] = (s, weakref.ref(self))
                off += n
        self._build_flip_table()
        self._seen = None
        self._dirty = True
        self._flips_dirty = False
        # fused optimisers update the weights without bumping their version counters, so any optimiser step marks
        # the working copies stale; plain in-place writes (load_state_dict, broadcast) are seen through the versions
        ref = weakref.ref(self)

        def _after_step(optimizer, args, kwargs):
            me = ref()
            if me is None:
                return
            if any(o is me for o in getattr(optimizer, "refreshed_working_sets", ())):
                me._flips_dirty = True      # ucd_amd.optim.SGD wrote the bf16 copies in its own pass: only the flipped set is stale
            else:
                me._dirty = True
        # frozen copies (the teacher) are not touched by any optimiser: no hook, the version counters (load_state_dict,
        # copy_) and mark_stale() are what invalidates them - otherwise every student step would re-cast the teacher
        self._hook = register_optimizer_step_post_hook(_after_step) if trainable else None
        self.refresh_if_stale()

    def __del__(self):
        hook = getattr(self, "_hook", None)
        if hook is not None:
            hook.remove()

    def _build_flip_table(self):
        """Flipped + transposed bf16 copies of the stride-1 weights whose input gradient runs as a FORWARD convolution
        (``blocks._StrideOneConvFn`` / the conv+ABN node: dx = conv2d(dy, w.flip(2, 3).transpose(0, 1)); for a 1x1 layer this is
        the transposed weight [Ci, Co] the own GEMM kernel takes): one flat buffer, refreshed together with
        the working copies by ONE batched kernel (``ucd_flip_weights_batched``) instead of a flip + copy per layer and step."""
        from .blocks import Conv1x1, Conv3x3
        self.flat16_flip = None
        if not self.trainable:
            return
        mods = [m for m in self.convs if isinstance(m, (Conv3x3, Conv1x1)) and m.weight.requires_grad and m.bias is None
                and m.stride == (1, 1) and m.groups == 1 and not (isinstance(m, Conv1x1) and m.as_gemm and not (m.own_dgrad or m.link_dgrad))
                and m.weight.is_contiguous(memory_format=torch.channels_last)]
        if not mods:
            return
        dev = self.flat16.device
        entries, blocks, off = [], [], 0
        base = self.flat16.data_ptr()
        for e, m in enumerate(mods):
            co, ci, kh, kw = m.weight.shape
            src_off = (m._w16.data_ptr() - base) // 2
            entries.append([src_off, off, co, ci, kh * kw])
            for sp in range(kh * kw):
                for a in range((co + 63) // 64):
                    for b in range((ci + 63) // 64):
                        blocks.append([e, sp, a, b])
            off += co * ci * kh * kw
        self.flat16_flip = torch.empty(off, dtype=torch.bfloat16, device=dev)
        self._flip_entries = torch.tensor(entries, dtype=torch.int64, device=dev)
        self._flip_blocks = torch.tensor(blocks, dtype=torch.int32, device=dev)
        off = 0
        for m in mods:
            co, ci, kh, kw = m.weight.shape
            n = co * ci * kh * kw
            # [ci, co, kh, kw] tensor in channels-last memory order ([ci][kh][kw][co])
            m._w16_flip = self.flat16_flip[off:off + n].view(ci, kh, kw, co).permute(0, 3, 1, 2)
            off += n

    def _refresh_flips(self):
        if self.flat16_flip is None:
            return
        from . import hip
        hip._check(hip.load().ucd_flip_weights_batched64(self.flat16.data_ptr(), self.flat16_flip.data_ptr(),
                                                         self._flip_blocks.data_ptr(), self._flip_blocks.shape[0],
                                                         self._flip_entries.data_ptr(), hip.stream()), "ucd_flip_weights_batched64")

    def mark_stale(self):
        self._dirty = True

    def refresh_if_stale(self):
        """One cast kernel for all weights, only when some master changed (in-place updates bump the version)."""
        if self.flat32 is None:
            return
        # ``w.data = view`` keeps each Parameter's own version counter, so the flat buffer's counter does not see the
        # optimiser's in-place updates; every step touches all weights, so the first and last stand for the set
        v = (self.convs[0].weight._version, self.convs[-1].weight._version)
        if self._dirty or v != self._seen:
            with torch.no_grad():
                self.flat16.copy_(self.flat32)
                self._refresh_flips()
            self._seen = v
            self._dirty = False
            self._flips_dirty = False
        elif self._flips_dirty:
            with torch.no_grad():
                self._refresh_flips()
            self._flips_dirty = False
