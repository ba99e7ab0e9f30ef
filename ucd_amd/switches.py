"""The ``UCD_*`` A/B switches of the product path, resolved ONCE per process.

Every switch is an environment variable read at its first use and cached in a plain dict (the layer code asks ~600 times per
step; ``os.environ.get`` encodes the key on every call).  Tests and probes that flip a switch inside one process go through
``set`` / ``unset`` (which also keep ``os.environ`` in step for child processes) or call ``reload`` after changing the
environment themselves.  ``snapshot`` is what bench.py prints as ``own_kernels``: a run that lost a kernel family to a switch
(or to a fallback) is visible in its JSON line.
"""
from __future__ import annotations

import os

# name -> default; "1" = the own kernel / fusion is on
DEFAULTS = {
    "UCD_FUSED_CONV1X1": "1",      # conv + training-ABN nodes on csrc/conv1x1.hip (0: module path, library kernels)
    "UCD_BWD_LINK": "1",           # backward links inside a bottleneck (out_mode 3)
    "UCD_BLOCK_LINK": "1",         # block links across a block boundary (out_mode 4)
    "UCD_PROJ_ALIAS": "1",         # projection blocks: proj_conv reads the alias returned by conv1's node
    "UCD_DGRAD_VIA_FWD": "1",      # stride-1 input gradients on forward solvers / the own kernel
    "UCD_LIB_GEMM_WIDE": "0",      # 1: the two large square 1x1 products on hipBLASLt
    "UCD_OWN3X3_MIN_TILES": "1",   # own 3x3 below this many tiles -> MIOpen
    "UCD_OWN3X3_WIDE": "1",        # 0: the 512 -> 512 3x3 layers on MIOpen
    "UCD_OWN_STRIDED": "1",        # strided conv2 / proj_conv on the own kernels
    "UCD_OWN_WGRAD": "1",          # weight gradients on csrc/wgrad.hip
    "UCD_OWN_HEADS": "1",          # classifier heads (Ct <= 64) on the own GEMM kernels through a zero-padded 64-row weight (0: library)
    "UCD_OWN_STEM": "1",           # 7x7/2 stem forward on csrc/stem.hip
    "UCD_WGRAD3": "1",             # 3x3 weight gradients: one kernel row per workgroup (0: the 9-tap form; read by the library)
    "UCD_CONV_PIPE": "auto",       # pipeline of the GEMM kernel: auto | 2x64 | 4x32 | 4x64 | lw32 | lw64 | lw256 (read by the library)
    "UCD_CONV_BN64_TILES": "128",  # launches of at most this many 128 x 128 tiles run on 128 x 64 tiles (0: never; read by the library)
    "UCD_CONV_LW64_TILES": "128",  # round 6: launches of at most this many 128 x 64 tiles run on 64-row loader-wave tiles - twice the CUs at 3 images per GPU (0: never; read by the library)
    "UCD_WGRAD_DEFER": "1",        # round 6: the slab sum of a weight gradient rides in the NEXT weight-gradient launch (0: a launch of its own behind every product)
    "UCD_WGRAD_STREAM": "1",       # round 6: the nodes' weight gradients run on a side stream of the library between the wrapper's flushes - off the chain of input-gradient products (0: on the compute stream).  UCD_WGRAD_STREAM_LATE (1: launches one call behind their fork point), UCD_WGRAD_STREAM_GROUP (1: calls per fork point) and UCD_WGRAD_STREAM_PRIO (low | normal) are read by the library
    "UCD_CONV3_MIN_ROWS": "0",     # stand-alone 3x3 layers (ASPP) below this many rows stay on the library path (round 6: 0 - the own kernels win at 3 / 6 images too: 9.33 -> 9.03 / 12.46 -> 11.97 ms)
    "UCD_STAT_ATOMIC": "1",        # conv + ABN nodes: statistics / link sums by fp32 atomics into arena slots, finalised by the apply passes (0: per-tile rows + reduction launches, bit-reproducible)
    "UCD_SEG_PK": "1",             # fused logit losses: packed math, fp64 LDS accumulators (0: the round-3 register form; read by the library)
    "UCD_STEM_EVAL_FUSED": "1",    # frozen-statistics stem (the teacher): conv1 + norm + activation + max pool as one kernel (0: two kernels)
    "UCD_STEM_FOLD": "1",          # stem norm + max-pool as one pass
    "UCD_ABN_NODE": "1",           # C++ autograd nodes
    "UCD_ABN_GENERIC": "0",        # 1: per-element ABN apply / backward kernels instead of the packed-pair forms (read by the library)
    "UCD_SGD": "hip",              # one-launch optimiser step (torch: torch's fused SGD)
    "UCD_STEP_GRAPH": "auto",      # whole-step hipGraph: auto = world 1 only, 1 = always try, 0 = never
    "UCD_TEACHER_OVERLAP": "1",    # frozen teacher on a side stream beside the student's forward (0: in front of it, same stream)
    "UCD_DIRECT_RCCL": "1",        # library-owned RCCL communicator for SyncBN
    "UCD_IPC_SYNC": "auto",        # SyncBN exchanges (<= 128 KB) as a one-shot IPC mailbox kernel: auto = only on an explicit attach_mailbox() (bench.py's third phase; every rank owns its GPU, self-test passed on all of them) - a plain run.py job stays on RCCL; 1 = always (also ranks sharing a GPU: tests); 0 = RCCL
    "UCD_IPC_TIMEOUT_MS": "60000", # give-up time of a mailbox exchange (a dead peer: the next collective raises instead of hanging; rank skew of a first iteration must fit)
    "UCD_DDP_LATE_COPY": "1",      # world 1: the bf16 -> fp32 gradient copies of all buckets as ONE launch at the end of the backward
    "UCD_DDP_DIRECT": "auto",      # gradient buckets over a library-owned RCCL communicator: auto = one-rank groups and captured multi-rank steps, 1 / 0 force
    "UCD_FORCE_COLLECTIVES": "0",  # 1 | abn | ddp: a one-rank process group still issues every SyncBN / gradient collective (bench.py --force_dist)
}

_cache: dict = {}


def get(name: str, default: str | None = None) -> str:
    v = _cache.get(name)
    if v is None:
        if default is None:
            default = DEFAULTS.get(name, "")
        v = _cache[name] = os.environ.get(name, default)
    return v


def set(name: str, value) -> None:       # noqa: A001 - mirrors os.environ's vocabulary
    os.environ[name] = str(value)
    _cache[name] = str(value)


def unset(name: str) -> None:
    os.environ.pop(name, None)
    _cache.pop(name, None)


def reload() -> None:
    """Forget the cached values (after ``os.environ`` / ``monkeypatch.setenv`` changed a switch)."""
    _cache.clear()


def snapshot() -> dict:
    return {k: get(k) for k in DEFAULTS}
