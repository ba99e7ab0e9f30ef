"""Benchmark of the hot path: images/sec of one UCD training iteration (VOC 15-5 step 1, 513x513,
global batch 24) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = teacher forward (eval) + student forward/backward + UnbiasedCE + contrastive/100 + 10*UnbiasedKD
+ gradient all-reduce + SGD + PolyLR on a device-resident synthetic batch (train.py:95-151 of the
reference).  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline and cpu_baseline
definitions.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _seed_miopen_user_db():
    """MIOpen's solver search (torch.backends.cudnn.benchmark) takes ~1 minute of the warm-up on a fresh box.  Its results
    for this workload's convolutions, captured on an MI355X, are committed under ucd_amd/tuning/miopen (text find-db of the
    same MIOpen build); seeding a private user-db directory with them turns every search into a lookup.  A different MIOpen
    build or GPU simply ignores the files (they are keyed by build and architecture) and searches as before."""
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("UCD_NO_MIOPEN_SEED"):
        return
    src = os.path.join(ROOT, "ucd_amd", "tuning", "miopen")
    if not os.path.isdir(src):
        return
    import atexit
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="ucd_miopen_db_")
    for name in os.listdir(src):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    atexit.register(shutil.rmtree, dst, True)      # a private copy per process (ranks must not share a writable db): removed at exit


if __name__ == "__main__" or os.environ.get("UCD_MIOPEN_SEED"):
    _seed_miopen_user_db()      # before torch / MIOpen load; the tools that import this module keep their own environment

import torch
import torch.distributed as dist

PEAK_HBM_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TF = 157.3    # dense fp32 matrix peak
PEAK_F16_MFMA_TF = 2500.0   # dense fp16/bf16 matrix peak
STAGING_ROOF_GBS = 14000.0  # 256 CUs x ~26 B/clk x 2.1 GHz through the LDS-DMA / L1 path (measured: tools/probes/fill_rate.hip, profiles/r06_lw_probe.txt)
MFMA_CALLS = ("ucd_conv3x3", "ucd_conv3x3_wgrad")     # 9 K deep implicit GEMMs: priced in flop (hip.py _timed work)
E_PER_IMAGE_513 = 98.44e6   # ABN activation elements per image per pass at 513^2


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--global_batch", type=int, default=24)
    p.add_argument("--force_dist", nargs="?", const="1", default=None, choices=["1", "abn", "ddp"],
                   help="one process: a one-rank RCCL group with every SyncBN / gradient collective issued (the N > 1 code path on "
                        "one GPU); abn / ddp: only the SyncBN or only the gradient-bucket collectives")
    p.add_argument("--crop", type=int, default=513)
    p.add_argument("--opt_level", default="O1", choices=["O0", "O1"])
    p.add_argument("--task", default="15-5")
    p.add_argument("--dataset", default="voc")
    p.add_argument("--step", type=int, default=1)
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_kernel_timing", action="store_true")
    p.add_argument("--no_miopen_find", action="store_true", help="skip the MIOpen solver search (default: search; ~2 min of warm-up, 25 % faster convolutions)")
    p.add_argument("--pixcon_precision", default=None, choices=["f32", "f16"])
    # test hooks: run the N > 1 code path with several ranks on ONE GPU (RCCL refuses that; gloo carries CUDA tensors)
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)
    p.add_argument("--device", type=int, default=None, help=argparse.SUPPRESS)
    p.add_argument("--check_lockstep", action="store_true", help=argparse.SUPPRESS)   # test hook: ranks compare weights / statistics
    p.add_argument("--first_step_losses", action="store_true", help=argparse.SUPPRESS)   # test hook: losses of the very first iteration
    return p.parse_args()


def build(args, device, per_rank_batch, rank):
    from ucd_amd import argparser, synth, tasks
    from ucd_amd.ddp import DistributedDataParallel
    from ucd_amd.run import build_models, load_step_checkpoint, make_optimizer
    from ucd_amd.scheduler import PolyLR
    from ucd_amd.train import Trainer
    opts = argparser.get_argparser().parse_args([
        "--method", "UCD", "--task", args.task, "--dataset", args.dataset, "--step", str(args.step), "--lr", "0.001",
        "--batch_size", str(per_rank_batch), "--crop_size", str(args.crop), "--no_pretrained",
        "--opt_level", args.opt_level, "--norm_act", "iabn_sync"])
    opts = argparser.modify_command_options(opts)
    classes = tasks.get_per_task_classes(args.dataset, args.task, args.step)
    new_ids, _, _ = tasks.get_task_labels(args.dataset, args.task, args.step)
    torch.manual_seed(opts.random_seed)
    model, model_old = build_models(opts, device, classes)
    # fake step-(k-1) checkpoint: deterministic weights, keys prefixed like a DDP-saved file (run.py:37)
    state = {"module." + k: v for k, v in synth.fill_state_dict(
        {k: v.cpu() for k, v in model_old.state_dict().items()}, 42, calibrated=True).items()}
    optimizer = make_optimizer(opts, model)
    scheduler = PolyLR(optimizer, max_iters=30 * 2145 // max(1, args.global_batch), power=opts.lr_power)
    model = DistributedDataParallel(model, delay_allreduce=True,
                                    bf16_weights=getattr(opts, "opt_level", "O0") != "O0" and getattr(opts, "bf16_weights", True))
    load_step_checkpoint(opts, model, model_old, state, device)
    opts.pixcon_precision = args.pixcon_precision
    trainer = Trainer(model, model_old, device=device, opts=opts, classes=classes)
    images = synth.images(1234 + rank, per_rank_batch, args.crop).to(device).contiguous(memory_format=torch.channels_last)
    labels = synth.seg_labels(1234 + rank, per_rank_batch, args.crop, args.crop, new_ids).to(device)
    model.train()
    return trainer, optimizer, scheduler, images, labels, classes


def kernel_timing(trainer, optimizer, scheduler, images, labels, steps):
    """Second, instrumented pass (not part of `value`): HIP events around every libucd_hip call on the
    stream it is launched on, plus the algorithmic bytes / flops of each call."""
    from ucd_amd import abn, blocks, hip
    # attribution mode: every library call visible and alone on the stream - no C++ nodes (their Python twins issue the
    # same library calls), no teacher graph, no teacher/student overlap (the timed region above ran with all three)
    saved = (abn._node_mod, trainer.graph_teacher, trainer._side, blocks._node_cache[0], trainer.step_graph)
    abn._abn_node()
    abn._node_mod, trainer.graph_teacher, trainer._side, trainer.step_graph = None, False, None, False
    blocks._node_cache[0] = None
    rec = hip.enable_call_timing()
    try:
        for _ in range(steps):
            trainer.train_step(images, labels, optimizer, scheduler)
        torch.cuda.synchronize()
    finally:
        hip.disable_call_timing()
        abn._node_mod, trainer.graph_teacher, trainer._side, blocks._node_cache[0], trainer.step_graph = saved
    # an event pair with nothing between it does not read zero (~5 us here); it is REPORTED, not subtracted: a call's
    # events bracket launch gaps too, so the per-call durations are upper bounds of the kernel time (rocprofv3's kernel
    # durations in profiles/ are ~10 % shorter) and the roofline fraction computed from them is conservative
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a, b in pairs:
        a.record(); b.record()
    torch.cuda.synchronize()
    overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
    out = {}
    for name, calls in rec.items():
        per_call = [c[0].elapsed_time(c[1]) for c in calls]
        ms = sum(per_call)
        work = sum(c[2] for c in calls)
        out[name] = {"launches": len(calls), "ms_total": ms, "avg_us": 1e3 * ms / max(1, len(calls)), "work": work}
        if calls and len(calls[0]) > 3:
            # per-SHAPE bound: a 1x1 product is priced against whichever of its two roofs is the longer -
            # bytes / 8 TB/s or flop / 2.5 PFLOP/s (2048 -> 512, 512 -> 2048 and 1024 -> 2048 are MFMA-bound, K N / (K + N) > 312)
            mix = {"hbm": [0.0, 0.0], "mfma": [0.0, 0.0]}      # measured ms, roof ms
            for t, c in zip(per_call, calls):
                t_h, t_m = c[2] / (PEAK_HBM_GBS * 1e9) * 1e3, c[3] / (PEAK_F16_MFMA_TF * 1e12) * 1e3
                k = "mfma" if t_m > t_h else "hbm"
                mix[k][0] += t
                mix[k][1] += max(t_h, t_m)
            out[name]["bound_mix"] = {"hbm_ms": mix["hbm"][0], "mfma_ms": mix["mfma"][0],
                                      "hbm_frac": mix["hbm"][1] / mix["hbm"][0] if mix["hbm"][0] else None,
                                      "mfma_frac": mix["mfma"][1] / mix["mfma"][0] if mix["mfma"][0] else None}
            if len(calls[0]) > 4:
                # round 6: the third roof of these products - the CUs' staging path (LDS-DMA: ~26 B/clk/CU measured, tools/probes/
                # fill_rate.hip; 256 CUs x 26 B/clk x 2.1 GHz = 14 TB/s).  The K >> N layers sit on it, below the HBM roof they are
                # priced against above; reported, not used for `frac`
                staged = sum(c[4] for c in calls)
                out[name]["staging"] = {"bytes": staged, "achieved_gbs": staged / (ms * 1e-3) / 1e9, "roof_gbs": STAGING_ROOF_GBS,
                                        "frac": staged / (ms * 1e-3) / 1e9 / STAGING_ROOF_GBS}
    out["_event_pair_overhead_us"] = 1e3 * overhead_ms
    return out


def cpu_baseline(args, classes):
    """The oracle (a port of the reference's CPU path) timed on this host's cores: ONE full UCD step on a
    2-image sample of the same workload."""
    from oracle import step as OS
    from oracle.params import student_teacher_params
    from ucd_amd import synth, tasks
    B = 8
    # measured on the GPU box (tools/cpu_thread_probe.py): the oracle's convolutions are fastest at 16
    # threads (0.32 s teacher forward) and 7x slower at 128 - more threads only add contention
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Ps, Pt = student_teacher_params(classes)
    new_ids, _, _ = tasks.get_task_labels(args.dataset, args.task, args.step)
    img = synth.images(1234, B, args.crop)
    lab = synth.seg_labels(1234, B, args.crop, args.crop, new_ids)
    params = [v for k, v in Ps.items() if v.requires_grad and not k.startswith("cls.0.")]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
    dt = None
    for _ in range(2):      # first pass warms the allocator / oneDNN primitives, second is reported
        opt.zero_grad()
        t0 = time.time()
        r = OS.ucd_losses(Ps, Pt, img, lab, classes)
        (r["loss"] + r["lkd"]).backward()
        opt.step()
        dt = time.time() - t0
    # the contrastive prep + loss alone (forward + backward) at the per-rank shape of the 8-GPU configs
    # (3 images, 33 x 33 maps, K = 16: A ~ 2.4k anchors x C ~ 4.5k contrast rows), same host threads
    from oracle import contrastive as OC
    f_n, f_o, l_po, lab3 = synth.contrastive_case(7, 3, 256, 33, 33, 16, args.crop, args.crop, new_ids)
    tc = None
    for _ in range(2):
        x = f_n.clone().requires_grad_(True)
        t0 = time.time()
        prep = OC.pre_contrastive_pixel(x, lab3, l_po, f_o)
        OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07).backward()
        tc = time.time() - t0
    A3, C3 = prep["a"].shape[0], prep["c"].shape[0]
    contrastive = {"seconds": tc, "A": A3, "C": C3, "K": 16, "gflops_algorithmic": A3 * C3 * (4 * 256 + 2 * 16) / tc / 1e9,
                   "sample": "pre_contrastive_pixel + PixelConLossV2 forward + backward, 3 images 33x33 maps (cfg3 per-rank shape)"}
    return {"value": B / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "contrastive": contrastive,
            "sample": f"1 full UCD step after 1 warm-up step (teacher fwd + student fwd/bwd + CE + contrastive + KD + SGD), {B} images "
                      f"{args.crop}x{args.crop}, fp32 PyTorch-CPU oracle, {dt:.1f} s",
            "loss": float((r["loss"] + r["lkd"]).detach())}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    device = torch.device("cuda", local_rank if args.device is None else args.device)
    torch.cuda.set_device(device)
    if args.force_dist and world == 1:
        # the multi-rank code path on ONE GPU: a one-rank RCCL process group, and every SyncBN all-gather / all-reduce and every
        # gradient bucket's all-reduce issued anyway (UCD_FORCE_COLLECTIVES, read by ucd_amd/abn.py and ucd_amd/ddp.py when they
        # are imported in build()) - what a box with one GPU can exercise of the N > 1 step, step graph included
        os.environ["UCD_FORCE_COLLECTIVES"] = args.force_dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29741")
        dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)
    assert args.global_batch % world == 0
    per_rank = args.global_batch // world
    torch.backends.cudnn.benchmark = not args.no_miopen_find

    trainer, optimizer, scheduler, images, labels, classes = build(args, device, per_rank, rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_warm):
        """n_warm untimed iterations, then EXACTLY args.steps timed ones between barrier + synchronize pairs; max over the ranks"""
        for _ in range(n_warm):
            trainer.train_step(images, labels, optimizer, scheduler)
        sync()
        before = trainer.graph_steps
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step(images, labels, optimizer, scheduler)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, trainer.graph_steps - before == args.steps

    from ucd_amd import switches
    # one-time work never belongs in the timed region, whatever W is: MIOpen's solver search and the GEMM tuning run in
    # the first step, the whole-step hipGraph is captured at the fourth call (the teacher's own graph at the third / fifth
    # when the step is not captured), the library-owned communicator is created at the first SyncBN layer - so at least six
    # untimed steps run before the clock starts
    multi = world > 1 or bool(args.force_dist)
    eager_ms = graph_ms = None
    # the one-shot mailbox exchange (csrc/comm.hip) is measured LAST and under a watchdog (mailbox_phase below): the eager and the
    # captured iterations are timed with the SyncBN exchanges on RCCL first, so a mailbox that misbehaves on a node this build has
    # never seen cannot cost the run its number
    mail_sw = switches.get("UCD_IPC_SYNC", "auto")
    if world > 1:
        switches.set("UCD_IPC_SYNC", "0")
    first_losses = None
    if args.first_step_losses:
        # the losses of iteration 1 depend on the initial weights alone: two equivalent arithmetics (plain / forced collectives,
        # eager / captured) are compared THERE, before nine optimiser steps of a chaotic network separate them
        trainer.train_step(images, labels, optimizer, scheduler)
        torch.cuda.synchronize()
        first_losses = {k: float(v) for k, v in trainer.last.items()}
    if not multi:
        dt, graphed = timed(max(args.warmup, 6))
    else:
        # More than one rank (or the forced-collectives mode): a captured multi-rank iteration has the 212 SyncBN exchanges and the
        # gradient buckets inside the graph - measured 11.2 against 14.9 ms eager at 3 images with one-rank collectives, never yet
        # on several GPUs, and the one failure mode seen on the way was a crash, not an exception.  So the EAGER iterations are
        # timed first and their result leaves the process (stderr + a file) before any capture is attempted; then the step is
        # captured (UCD_STEP_GRAPH != 0, both collective kinds on library-owned communicators, all ranks agreeing) and timed again.
        # `value` is the better of the two; execution.eager_ms / graph_ms carry both.
        want_graph = switches.get("UCD_STEP_GRAPH", "auto") != "0"
        trainer.step_graph = False
        dt, graphed = timed(max(args.warmup, 6))
        eager_ms = 1e3 * dt / args.steps
        if rank == 0:
            note = {"phase": "eager (before any capture)", "n_gpus": world, "ms_per_step": eager_ms,
                    "value": args.global_batch * args.steps / dt, "unit": "images/sec", "global_batch": args.global_batch,
                    "forced_collectives": bool(args.force_dist and world == 1)}
            print("UCD_BENCH_EAGER " + json.dumps(note), file=sys.stderr, flush=True)
            try:
                with open(os.environ.get("UCD_BENCH_EAGER_FILE", os.path.join(ROOT, f"bench_eager_n{world}.json")), "w") as f:
                    json.dump(note, f)
            except OSError:
                pass
        if want_graph and trainer.enable_multi_rank_step_graph():
            dt2, graphed2 = timed(trainer.step_graph_warmup + 3)
            if graphed2:
                graph_ms = 1e3 * dt2 / args.steps
                if dt2 < dt:
                    dt, graphed = dt2, True
    last = {k: float(v) for k, v in trainer.last.items()}
    # device memory of this rank up to the end of the timed phases (the instrumented pass and the CPU baseline come later): the
    # captured step's pool included; the weight gradients' side stream keeps every layer's dZ until its join
    memory = {"peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
              "peak_reserved_gb": round(torch.cuda.max_memory_reserved(device) / 2 ** 30, 2)}
    # how the timed iterations actually ran: a silent fallback (capture failed, a kernel family switched off) is visible here
    execution = {"step_graph": bool(graphed),
                 "step_graph_error": trainer.step_graph_error,
                 "eager_ms": eager_ms, "graph_ms": graph_ms,      # multi-rank runs: both phases (None: not run / not captured)
                 # which phase `value` / `ms_per_step` were taken from (multi-rank runs time up to three; the best valid one counts)
                 "value_phase": ("graph" if graphed else "eager") if multi else ("graph" if graphed else "eager"),
                 "teacher_graph": bool(trainer._sg is not None or trainer._tg is not None),
                 "teacher_graph_error": getattr(trainer, "teacher_graph_error", None),
                 "teacher_overlap": trainer._side is not None,
                 # --force_dist: a one-rank RCCL group with every SyncBN / gradient collective issued (NOT a multi-GPU number)
                 "forced_collectives": bool(args.force_dist and world == 1)}
    own_kernels = switches.snapshot()
    own_kernels["UCD_IPC_SYNC"] = mail_sw       # as the caller set it (execution.mailbox says what the third phase did with it)
    def check_lockstep():
        # data parallelism keeps every replica identical: after the timed steps all ranks must hold the same parameters AND
        # the same running statistics (SyncBN), bit for bit (gradients are averaged before the update, statistics are
        # combined from the same gathered table on every rank)
        net = trainer.model.module if hasattr(trainer.model, "module") else trainer.model
        sig = torch.stack([t.detach().double().sum() for t in list(net.parameters()) + list(net.buffers())
                           if t.is_floating_point()])
        gathered = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(gathered, sig)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        if not same and rank == 0:
            names = [n for n, t in list(net.named_parameters()) + list(net.named_buffers()) if t.is_floating_point()]
            bad = [(names[i], [float(g[i] - gathered[0][i]) for g in gathered]) for i in range(len(names))
                   if any(g[i] != gathered[0][i] for g in gathered)]
            print("UCD_BENCH_LOCKSTEP %d of %d tensors differ between the ranks; first: %s" % (len(bad), len(names), bad[:6]),
                  file=sys.stderr, flush=True)
        return same

    roof, kernels = None, None
    if not args.no_kernel_timing:
        kernels = kernel_timing(trainer, optimizer, scheduler, images, labels, min(args.steps, 3))
        ev_us = kernels.pop("_event_pair_overhead_us")
        for kname, kv in kernels.items():                    # per-call-type roofline fractions
            secs = max(kv["ms_total"] * 1e-3, 1e-12)
            if kname.startswith("ucd_pixcon_loss") or kname in MFMA_CALLS:
                pk = PEAK_F32_MFMA_TF if "f32" in kname else PEAK_F16_MFMA_TF      # bf16 and fp16 share the dense peak
                kv.update(bound="mfma", achieved=kv["work"] / secs / 1e12, unit="TFLOP/s", frac=kv["work"] / secs / 1e12 / pk)
            else:
                kv.update(bound="hbm", achieved=kv["work"] / secs / 1e9, unit="GB/s", frac=kv["work"] / secs / 1e9 / PEAK_HBM_GBS)
        name = max(kernels, key=lambda k: kernels[k]["ms_total"])
        k = kernels[name]
        if name.startswith("ucd_pixcon_loss") or name in MFMA_CALLS:
            ach = k["work"] / (k["ms_total"] * 1e-3) / 1e12
            peak = PEAK_F32_MFMA_TF if "f32" in name else PEAK_F16_MFMA_TF
            roof = {"kernel": name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": None}
        else:
            ach = k["work"] / (k["ms_total"] * 1e-3) / 1e9
            roof = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": None}
        if "staging" in k:
            roof["staging"] = k["staging"]           # the same launches against the CUs' LDS staging rate (a roof below HBM for K >> N)
        if "bound_mix" in k:
            roof["bound_mix"] = k["bound_mix"]       # the call type split by each call's own roof (ms in the instrumented pass, fractions)
        roof["avg_launch_us"] = k["avg_us"]
        roof["event_pair_overhead_us"] = ev_us       # empty event pair, for reference (not subtracted)
        roof["launches_per_step"] = k["launches"] / min(args.steps, 3)
        # HBM traffic of that kernel from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # --pmc WRITE_SIZE, separate runs, gfx950 corrections applied by tools/pmc_to_json.py); null if absent
        for rnd in ("r06", "r05", "r04", "r03", "r02"):     # the newest committed collection that has this call
            pmc = os.path.join(ROOT, "profiles", f"{rnd}_pmc_bench.json")
            if not os.path.exists(pmc):
                continue
            try:
                rec = json.load(open(pmc)).get(name.split("[")[0])
                if rec and rec.get("global_batch") == args.global_batch and world == 1:
                    roof["traffic"] = rec["bytes_per_launch"]
                    roof["traffic_source"] = f"profiles/{rnd}_pmc_bench.json"
                    break
            except (OSError, ValueError):
                pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args, classes)
        except Exception as e:  # the baseline is a reported extra: never lose the GPU measurement over it
            cpu = {"error": repr(e)[:200]}

    if rank == 0:
        out = {
            "metric": "images/sec (whole node) per UCD train step, VOC 15-5 step-1, 513^2 bs=24",
            "value": args.global_batch * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if args.opt_level != "O0" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.dataset} {args.task} step-{args.step} --method UCD, ResNet-101/DeepLab-V3, "
                                   f"{args.crop}x{args.crop}, global batch {args.global_batch} "
                                   f"({per_rank}/GPU), random-init weights via a synthetic step-0 checkpoint",
                       "parallelism": f"dp{world}", "opt_level": args.opt_level,
                       "contrastive_dtype": trainer.pixcon_precision},
            "losses": last, "execution": execution, "memory": memory, "own_kernels": own_kernels,
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
            # what the per-call table (`kernels`, `roofline.achieved`) was measured on: NOT the timed configuration itself
            "kernel_table_config": ("instrumented eager pass after the timed region: HIP events around every C-ABI call, Python twins of "
                                    "the C++ autograd nodes on the deterministic statistics path (per-tile partial rows + reduction "
                                    "launches), no step graph, no teacher overlap; the timed region ran the C++ nodes with atomic "
                                    "statistics inside one hipGraph - profiles/*_step_kernel_summary*.txt (rocprofv3 of the same "
                                    "command) is the authority for its kernels") if kernels is not None else None,
        }
        if first_losses is not None:
            out["first_step_losses"] = first_losses
    else:
        out = None

    def emit():
        # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's stdio buffer until exit and would
        # otherwise land behind it (a multi-GPU line must stay parseable as "the last line")
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)

    if world > 1 and mail_sw != "0":
        # Third phase: the SyncBN exchanges on the mailbox.  `out` already holds the RCCL phases' result; a watchdog on every rank
        # prints it (rank 0) and leaves if this phase does not come back, an exception does the same at once.
        import threading
        limit = float(os.environ.get("UCD_BENCH_MAILBOX_LIMIT_S", "240"))
        finished = threading.Event()

        def bail(why):
            # the line of phases 1 - 2 still leaves the process (their measurements are complete and valid); the mailbox failure is in
            # execution.mailbox.error and on stderr.  Exit status: 0 by default (the driver reads the line of a run whose RCCL phases
            # succeeded), 3 with UCD_BENCH_STRICT=1 (CI: a failed or hung mailbox phase is a failed run - ADVICE r5)
            print("UCD_BENCH_MAILBOX_FAILED rank %d: %s" % (rank, why), file=sys.stderr, flush=True)
            if rank == 0:
                out["execution"]["mailbox"] = {"error": why}
                emit()
            sys.stderr.flush()
            os._exit(3 if os.environ.get("UCD_BENCH_STRICT") == "1" else 0)

        def watchdog():
            if not finished.wait(limit):
                bail("watchdog: the mailbox phase did not finish within %.0f s" % limit)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            from ucd_amd.comm import attach_mailbox, mailbox_timeouts
            switches.set("UCD_IPC_SYNC", mail_sw)
            if "UCD_IPC_TIMEOUT_MS" not in os.environ:
                switches.set("UCD_IPC_TIMEOUT_MS", "10000")
            info = {"attached": bool(attach_mailbox(None))}
            if info["attached"]:
                if not (graph_ms is not None and trainer.enable_multi_rank_step_graph()):
                    trainer.step_graph = False
                dt3, graphed3 = timed(trainer.step_graph_warmup + 3)
                t = torch.tensor([float(mailbox_timeouts(None))], device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                info.update(ms_per_step=1e3 * dt3 / args.steps, step_graph=bool(graphed3), timeouts=int(t.item()))
                # the mailbox phase counts only when no exchange timed out AND the ranks are still bit-identical replicas after it
                info["lockstep"] = bool(check_lockstep())
                info["finite"] = all(math.isfinite(float(v)) for v in trainer.last.values())
                if rank == 0 and info["timeouts"] == 0 and info["lockstep"] and info["finite"] and dt3 < dt:
                    out.update(value=args.global_batch * args.steps / dt3, ms_per_step=1e3 * dt3 / args.steps)
                    out["execution"]["step_graph"] = bool(graphed3)
                    out["execution"]["value_phase"] = "mailbox"
                    out["losses"] = {k: float(v) for k, v in trainer.last.items()}
            if rank == 0:
                out["execution"]["mailbox"] = info
        except Exception as e:
            bail("mailbox phase failed on rank %d: %s" % (rank, repr(e)[:300]))
        finished.set()
    if args.check_lockstep and world > 1:
        same = check_lockstep()
        if rank == 0:
            out["lockstep"] = bool(same)
    if world > 1 or (args.force_dist and dist.is_initialized()):
        dist.destroy_process_group()
    if rank == 0:
        emit()


if __name__ == "__main__":
    main()
