"""``Trainer``: one UCD training iteration on an MI355X, reference interface (train.py:15-384).

``Trainer(model, model_old, device, opts, trainer_state, classes)`` and
``.train(cur_epoch, optim, train_loader, scheduler, print_int, logger) -> (epoch_loss, reg_loss)`` keep
the reference's call shapes.  The iteration is the reference's ``train.py:95-151`` with the UCD branch
repaired as intended (``pre_contractive_pixel`` returns five values and
``PixelConLossV2`` takes five):

    teacher forward (eval, no grad)          train.py:100-102
    student forward (train)                  :108
    loss = mean(CE(out, labels)) + PixCon(student/teacher pre-logits) / 100          :115-116
    lkd  = loss_kd * KD(out, out_old)        :131-133
    backward; gradient all-reduce; SGD; PolyLR step                                   :135-151

What changed for the hardware: activations are channels-last bf16 (``--opt_level`` O1..O3) or fp32 (O0);
the contrastive term is one fused HIP operation with no host round trip; the per-iteration ``.item()``
reads (train.py:153-157, >= 5 device syncs) are replaced by device-side accumulators read once per
``print_int`` iterations; gradient averaging overlaps the backward (ucd_amd.ddp).
Out of scope on this path (raise ``NotImplementedError``): BCE/iCaRL losses and the EWC/RW/PI
regularisers of the other baselines.
"""
from __future__ import annotations

from functools import reduce

import torch
import torch.distributed as dist
import torch.nn as nn

from . import switches as _switches
from .contrastive import ucd_contrastive_loss
from .loss import (KnowledgeDistillationLoss, UnbiasedCrossEntropy, UnbiasedKnowledgeDistillationLoss,
                   fused_seg_losses)


def _raw(features, name):
    return features.raw(name) if hasattr(features, "raw") else features[name]


class Trainer:
    def __init__(self, model, model_old, device, opts, trainer_state=None, classes=None):
        self.model_old, self.model, self.device = model_old, model, device
        if classes is not None:
            tot_classes = reduce(lambda a, b: a + b, classes)
            self.old_classes = tot_classes - classes[-1]
            self.tot_classes = tot_classes
        else:
            self.old_classes, self.tot_classes = 0, getattr(opts, "num_classes", None) or 21
        if opts.bce or opts.icarl:
            raise NotImplementedError("BCE / iCaRL (--bce, --icarl, --method LWF-MC) are other baselines, "
                                      "outside the UCD hot path")
        if getattr(opts, "regularizer", None) is not None:
            raise NotImplementedError("EWC / RW / PI regularisers are outside the UCD hot path")
        self.temperature = opts.temperature
        self.pixcon_weight = getattr(opts, "pixcon_weight", 0.01)      # the reference hard-codes /100 (train.py:116)
        # the reference clamps down-sampled labels at the VOC bound 20 (utils/utils.py:267-268); datasets
        # with more classes need the real bound
        self.max_label = max(20, self.tot_classes - 1)
        if opts.unce and self.old_classes != 0:
            self.criterion = UnbiasedCrossEntropy(old_cl=self.old_classes, ignore_index=255, reduction="none")
        else:
            self.criterion = nn.CrossEntropyLoss(ignore_index=255, reduction="none")
        self.lde = opts.loss_de
        self.lde_flag = self.lde > 0. and model_old is not None
        self.lde_loss = nn.MSELoss()
        self.lkd = opts.loss_kd
        self.lkd_flag = self.lkd > 0. and model_old is not None
        self.lkd_loss = (UnbiasedKnowledgeDistillationLoss if opts.unkd else KnowledgeDistillationLoss)(alpha=opts.alpha)
        self.regularizer, self.regularizer_flag = None, False
        # The frozen teacher (eval mode, no gradients, no collectives) is replayed from a hipGraph after two eager
        # warm-up steps: ~330 kernel launches of host work per step disappear, which matters once the per-GPU batch
        # is small (8-GPU regime: the step is launch-bound, not GPU-bound).
        self.graph_teacher = bool(getattr(opts, "graph_teacher", True)) and device.type == "cuda" and model_old is not None
        self._tg = None
        self._tg_seen = 0
        # the teacher has no dependence on the student before the losses: it runs on a side stream next to the student's
        # forward, filling the gaps that the small 33x33 layers of either network leave on the GPU
        self.overlap_teacher = (bool(getattr(opts, "overlap_teacher", True)) and device.type == "cuda" and model_old is not None
                                and _switches.get("UCD_TEACHER_OVERLAP", "1") != "0")
        self._side = torch.cuda.Stream(device) if self.overlap_teacher else None
        self.ret_intermediate = self.lde
        self.unce = bool(opts.unce and self.old_classes != 0)
        # fused up-sampling + CE + KD kernel whenever the loss pair is one it implements:
        # (unbiased or plain) CE, optionally with the unbiased KD
        # (the kernel has no --alpha: utils/loss.py:158 scales the teacher logits by it, so alpha != 1 takes the unfused path)
        self.fuse_logit_losses = (getattr(opts, "fused_logit_losses", True) and device.type == "cuda"
                                  and (not self.lkd_flag or (opts.unkd and self.unce and float(opts.alpha) == 1.0)))
        self.amp = getattr(opts, "opt_level", "O0") != "O0"
        # contrastive arithmetic: exact fp32 MFMA with fp32 activations (O0), fp16 operands otherwise
        self.pixcon_precision = getattr(opts, "pixcon_precision", None) or ("f16" if self.amp else "f32")
        # frozen teacher under autocast: static bf16 copies of its convolution weights (no per-call casts)
        self._teacher_w16 = None
        if self.amp and device.type == "cuda" and model_old is not None and getattr(opts, "bf16_weights", True):
            from .master import Bf16Weights
            self._teacher_w16 = Bf16Weights(model_old, trainable=False)
        self.last = {}
        # The WHOLE iteration (teacher + student forward, losses, backward, gradient buckets, optimiser) as one hipGraph,
        # captured after the eager warm-up iterations and replayed from then on: at the per-rank batch of the 8-GPU run
        # (3 images) the host needs ~14 ms to enqueue ~1200 launches of ~12 ms of kernels; a replay needs none of it.
        # UCD_STEP_GRAPH = auto (default): single-process runs only - a multi-rank capture would put the SyncBN and gradient
        # collectives into the graph, which no box available to this build can exercise; 1 = always try; 0 = never.
        sw = _switches.get("UCD_STEP_GRAPH", "auto")
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.step_graph = (device.type == "cuda" and bool(getattr(opts, "step_graph", True)) and sw != "0"
                           and (world == 1 or sw == "1") and not self.lde_flag)
        self.step_graph_warmup = 3          # eager iterations before the capture (solver search, GEMM tuning, optimiser tables)
        self._sg = None
        self._sg_seen = 0
        self.graph_steps = 0                # iterations served by a graph replay (bench.py reports it)
        self.step_graph_error = None

    def enable_multi_rank_step_graph(self):
        """Turn the captured iteration on for a run with more than one rank (or a one-rank group with forced collectives), AFTER the
        eager iterations have been measured: both collective kinds must sit on library-owned RCCL communicators (the SyncBN exchanges:
        ucd_amd.comm.direct_comm of the default group; the gradient buckets: GradReducer.enable_direct) and every rank must agree -
        otherwise the run stays eager.  Collective: every rank calls it at the same point.  Returns True when the next iterations
        will be captured."""
        if self.device.type != "cuda" or _switches.get("UCD_STEP_GRAPH", "auto") == "0" or self.lde_flag:
            return False
        ok = True
        if dist.is_available() and dist.is_initialized():
            from . import abn as _abn
            from .comm import _agree, direct_comm
            reducer = getattr(self.model, "reducer", None)
            ok = reducer is not None and reducer.enable_direct()
            ok = _agree(ok, None, self.device)
            if ok and (dist.get_world_size() > 1 or _abn._FORCE_SYNC):
                ok = direct_comm(None) is not None                          # same answer on every rank (its own agreement rounds)
        if ok:
            self.step_graph, self._sg, self._sg_seen, self.step_graph_error = True, None, 0, None
        return ok

    # ------------------------------------------------------------------------------------------
    def _autocast(self):
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.amp and self.device.type == "cuda")

    def _teacher_eager(self, images, up):
        with torch.no_grad(), self._autocast():
            return self.model_old(images, x_b_old=None, x_pl_old=None, ret_intermediate=self.ret_intermediate, **up)

    def _teacher_forward(self, images, up):
        """(outputs_old, features_old); replayed from a captured graph when enabled and the input shape is stable."""
        if self._teacher_w16 is not None:
            self._teacher_w16.refresh_if_stale()
        if not self.graph_teacher or self.lde_flag or torch.cuda.is_current_stream_capturing():
            return self._teacher_eager(images, up)          # (inside a whole-step capture the teacher is part of that graph)
        tg = self._tg
        if tg is not None and tg["shape"] == tuple(images.shape) and tg["up"] == bool(up):
            tg["images"].copy_(images)
            tg["graph"].replay()
            return tg["out"]
        self._tg_seen += 1
        if self._tg_seen <= 2:                      # eager warm-up: MIOpen solver search, workspaces, constants
            return self._teacher_eager(images, up)
        if self.step_graph and self._sg is None and self._sg_seen <= self.step_graph_warmup + 1:
            return self._teacher_eager(images, up)  # the whole-step capture is about to include the teacher: no graph of its own
        try:
            from .segmentation_module import Features
            static = images.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL / watchdog threads of a multi-GPU run keep making HIP calls while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                outputs_old, f = self._teacher_eager(static, up)
                # materialise exactly what the step reads; the lazy attention maps are not part of the graph
                out = (outputs_old, Features(f.raw("body"), f.raw("pre_logits"), f["sem"]))
            self._tg = {"graph": graph, "images": static, "out": out, "shape": tuple(images.shape), "up": bool(up)}
            graph.replay()
            return out
        except Exception as e:                      # capture is an optimisation: never lose the step over it
            self.graph_teacher = False
            self._tg = None
            self.teacher_graph_error = repr(e)[:200]            # bench.py prints it: a run without the graph is marked
            torch.cuda.synchronize()
            import warnings
            warnings.warn(f"teacher graph capture disabled: {e!r}")
            return self._teacher_eager(images, up)

    # -- the whole iteration as one graph --------------------------------------------------------------------------------
    def _graph_ready(self, optim):
        """Static preconditions of a capture: the gradient-bucket wrapper (gradients live at fixed addresses) and the
        one-launch optimiser with its tables in place (its hyper-parameters can then live on the device)."""
        from .optim import SGD
        return (hasattr(self.model, "finish_grad_sync") and hasattr(self.model, "zero_grad") and isinstance(optim, SGD)
                and optim.plan_is_current())

    def _graph_step(self, images, labels, optim, scheduler):
        """Replay (or capture, then replay) the iteration; None when this call has to run eagerly."""
        key = (tuple(images.shape), tuple(labels.shape), images.dtype, labels.dtype, id(optim), self.model.training)
        sg = self._sg
        if sg is not None and sg["key"] == key and not optim.plan_is_current():
            # parameters / gradients / momentum buffers moved (load_state_dict, add_param_group, a re-wrapped model): the captured
            # optimiser launch holds the old addresses - drop the graph, run eagerly, capture again after the warm-up count
            self._sg, self._sg_seen, sg = None, 0, None
            optim.device_hyper(False)
        if sg is not None and sg["key"] == key:
            sg["images"].copy_(images, non_blocking=True)
            sg["labels"].copy_(labels, non_blocking=True)
            optim.push_hyper()
            sg["graph"].replay()
            bw = getattr(self.model, "bf16_weights", None)
            if bw is not None:
                # the replayed optimiser step staled the flipped / transposed weights for any EAGER iteration that follows (another
                # batch shape, a dropped graph): the optimiser's post-hook, which sets this flag, does not run in a replay
                bw._flips_dirty = True
            if scheduler is not None:
                scheduler.step()
            self.graph_steps += 1
            self.last = sg["out"]
            return self.last
        if sg is not None:                          # another batch shape (last batch of an epoch, validation crop): eager
            return None
        self._sg_seen += 1
        if self._sg_seen <= self.step_graph_warmup or not self._graph_ready(optim):
            return None
        dev = self.device
        try:
            static_images = images.to(dev, dtype=torch.float32).contiguous(memory_format=torch.channels_last).clone(
                memory_format=torch.preserve_format)
            static_labels = labels.to(dev, dtype=torch.long).clone()
            optim.device_hyper(True)
            bw = getattr(self.model, "bf16_weights", None)
            if bw is not None:
                bw._flips_dirty = True              # the replayed iteration refreshes the flipped weights its optimiser step staled
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self._eager_step(static_images, static_labels, optim, None)
            self._sg = {"graph": graph, "images": static_images, "labels": static_labels, "out": out, "key": key}
            self._tg = None                         # the separate teacher graph (and its memory pool) is not needed any more
        except Exception as e:                      # capture is an optimisation: never lose the step over it
            self.step_graph = False
            self._sg = None
            self.step_graph_error = repr(e)[:300]
            optim.device_hyper(False)
            torch.cuda.synchronize()
            import warnings
            warnings.warn(f"whole-step graph capture disabled: {e!r}")
            if hasattr(self.model, "zero_grad"):
                self.model.zero_grad()
            return None
        return self._graph_step(images, labels, optim, scheduler)        # nothing ran during the capture: replay it now

    def train_step(self, images, labels, optim, scheduler=None):
        """One iteration; returns device scalars (no host synchronisation)."""
        if self.step_graph and self.model_old is not None and self.model.training:
            out = self._graph_step(images, labels, optim, scheduler)
            if out is not None:
                return out
        return self._eager_step(images, labels, optim, scheduler)

    def _eager_step(self, images, labels, optim, scheduler=None):
        model, model_old = self.model, self.model_old
        images = images.to(self.device, dtype=torch.float32, non_blocking=True)
        labels = labels.to(self.device, dtype=torch.long, non_blocking=True)
        if self.device.type == "cuda":
            images = images.contiguous(memory_format=torch.channels_last)
        zero = torch.zeros((), device=self.device)
        lkd = lde = zero
        fuse = self.fuse_logit_losses
        up = {} if not fuse else {"upsample": False}
        if model_old is not None:
            if self._side is not None:
                main = torch.cuda.current_stream(self.device)
                self._side.wait_stream(main)                      # the batch is ready; last step's readers are done
                with torch.cuda.stream(self._side):
                    outputs_old, features_old = self._teacher_forward(images, up)
            else:
                outputs_old, features_old = self._teacher_forward(images, up)
        if hasattr(model, "zero_grad") and hasattr(model, "finish_grad_sync"):
            model.zero_grad()
        else:
            optim.zero_grad(set_to_none=True)
        with self._autocast():
            if model_old is None:
                outputs, features = model(images, ret_intermediate=self.ret_intermediate, **up)
            else:
                outputs, features = model(images, x_b_old=_raw(features_old, "body"),
                                          x_pl_old=_raw(features_old, "pre_logits"),
                                          ret_intermediate=self.ret_intermediate, **up)
        if model_old is not None and self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)     # teacher outputs are needed from here on
        if fuse:
            # one pass over the label map: bilinear up-sampling + CE (+ KD) + gradient w.r.t. the low-res logits
            total, ce, kd = fused_seg_losses(features["sem"], features_old["sem"] if self.lkd_flag else None, labels,
                                             self.old_classes if self.unce else 1, 1.0,
                                             self.lkd if self.lkd_flag else 0.0)
        else:
            ce = self.criterion(outputs.float() if outputs.dtype != torch.float32 else outputs, labels).mean()
        con = zero
        if model_old is not None and self.pixcon_weight != 0:
            con = ucd_contrastive_loss(_raw(features, "pre_logits"), labels, features_old["sem"],
                                       _raw(features_old, "pre_logits"), self.temperature, self.max_label,
                                       self.pixcon_precision)
        loss = ce + con * self.pixcon_weight                                      # train.py:116 (/100)
        if self.lde_flag:
            lde = self.lde * (self.lde_loss(features["body"].float(), features_old["body"].float())
                              + self.lde_loss(features["pre_logits"].float(), features_old["pre_logits"].float()))
        if self.lkd_flag:
            lkd = self.lkd * (kd if fuse else self.lkd_loss(outputs, outputs_old))   # train.py:131-133
        loss_tot = (total + con * self.pixcon_weight + lde) if fuse else (loss + lkd + lde)
        loss_tot.backward()
        if hasattr(model, "finish_grad_sync"):
            model.finish_grad_sync()
        optim.step()
        if scheduler is not None:
            scheduler.step()
        self.last = {"loss": loss.detach(), "lkd": lkd.detach(), "lde": lde.detach(), "ce": ce.detach(),
                     "con": con.detach()}
        return self.last

    def train(self, cur_epoch, optim, train_loader, scheduler=None, print_int=10, logger=None):
        """Train one epoch and return (epoch_loss, reg_loss) like the reference (train.py:76-183)."""
        if logger is not None:
            logger.info("Epoch %d, lr = %f" % (cur_epoch, optim.param_groups[0]["lr"]))
        dev = self.device
        epoch_loss = torch.zeros((), device=dev)
        reg_loss = torch.zeros((), device=dev)
        interval = torch.zeros((), device=dev)
        if hasattr(train_loader, "sampler") and hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(cur_epoch)
        self.model.train()
        n = 0
        for cur_step, (images, labels) in enumerate(train_loader):
            r = self.train_step(images, labels, optim, scheduler)
            epoch_loss += r["loss"]
            reg_loss += r["lkd"] + r["lde"]
            interval += r["loss"] + r["lkd"] + r["lde"]
            n += 1
            if (cur_step + 1) % print_int == 0:
                value = (interval / print_int).item()           # the only host sync of the interval
                self._check_mailbox()                           # a timed-out SyncBN mailbox exchange (its results are NaN) raises here
                if logger is not None:
                    logger.info(f"Epoch {cur_epoch}, Batch {cur_step + 1}/{len(train_loader)}, Loss={value}")
                    logger.add_scalar("Loss", value, cur_epoch * len(train_loader) + cur_step + 1)
                interval.zero_()
        if dist.is_available() and dist.is_initialized():
            if n:
                torch.cuda.synchronize() if dev.type == "cuda" else None
                self._check_mailbox()                           # before the epoch's reductions: a poisoned run must not report a loss
            dist.reduce(epoch_loss, dst=0)
            dist.reduce(reg_loss, dst=0)
            world = dist.get_world_size()
            rank = dist.get_rank()
        else:
            world, rank = 1, 0
        if rank == 0 and n:
            epoch_loss = epoch_loss / world / n
            reg_loss = reg_loss / world / n
        if logger is not None:
            logger.info(f"Epoch {cur_epoch}, Class Loss={epoch_loss}, Reg Loss={reg_loss}")
        return (epoch_loss, reg_loss)

    def _check_mailbox(self):
        """The one-shot SyncBN mailbox (csrc/comm.hip) latches a timed-out exchange in pinned host memory; a replayed step graph issues
        no host-side collective call that would report it, so the training loop polls the word at its host synchronisations and
        raises.  Costs one read of host memory; a no-op without a mailbox."""
        if self.device.type == "cuda" and dist.is_available() and dist.is_initialized():
            from .comm import check_mailbox
            check_mailbox(None)

    def validate(self, loader, metrics, ret_samples_ids=None, logger=None):
        """Evaluation loop (train.py:185-270): class loss + confusion matrix, accumulated on the device."""
        metrics.reset()
        model = self.model
        dev = self.device
        class_loss = torch.zeros((), device=dev)
        reg_loss = torch.zeros((), device=dev)
        model.eval()
        n = 0
        with torch.no_grad():
            for images, labels in loader:
                images = images.to(dev, dtype=torch.float32)
                labels = labels.to(dev, dtype=torch.long)
                if self.fuse_logit_losses and hasattr(metrics, "update_from_logits"):
                    # low-resolution logits only: loss and confusion matrix come from the fused kernels (the reference
                    # up-samples, copies predictions to the host and histograms there, train.py:236-246)
                    with self._autocast():
                        _, feats = model(images, ret_intermediate=False, upsample=False)
                    sem = feats["sem"]
                    class_loss += fused_seg_losses(sem, None, labels, self.old_classes if self.unce else 1, 1.0, 0.0)[1]
                    metrics.update_from_logits(labels, sem)
                else:
                    with self._autocast():
                        outputs, _ = model(images, ret_intermediate=False)
                    class_loss += self.criterion(outputs.float(), labels.clone()).mean()
                    metrics.update(labels, outputs.argmax(dim=1))
                n += 1
            metrics.synch(dev)
            score = metrics.get_results()
        if dist.is_available() and dist.is_initialized():
            dist.reduce(class_loss, dst=0)
            world = dist.get_world_size()
        else:
            world = 1
        class_loss = class_loss / world / max(n, 1)
        if logger is not None:
            logger.info(f"Validation, Class Loss={class_loss}, Reg Loss={reg_loss} (without scaling)")
        return (class_loss, reg_loss), score, []

    def state_dict(self):
        return {"regularizer": None}

    def load_state_dict(self, state):
        pass
