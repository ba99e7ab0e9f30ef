"""Library-owned RCCL communicators for the per-layer SyncBN exchanges (one process per GPU).

``torch.distributed`` stays the control plane (rendezvous, broadcasting the RCCL unique id, gradient buckets in
``ucd_amd.ddp``); the 212 tiny statistics collectives of a step go through ``libucd_hip.so`` instead, on the compute
stream and inside the layer's own library call (``ucd_abn_sync_forward_comm`` / ``_backward_comm``): no dispatcher,
no hop to c10d's communication stream and back.  ``direct_comm(group)`` returns ``None`` whenever that path is not
available (backend is not nccl, ``UCD_DIRECT_RCCL=0``, RCCL cannot be bound, or the self-test disagrees on any rank);
the caller then uses the ``torch.distributed`` collectives around the split library calls.
"""
from __future__ import annotations

import ctypes as C
import os

from . import switches as _switches
import warnings

import torch
import torch.distributed as dist

from . import hip

_comms = {}


class DirectComm:
    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, world, rank


def _loaded_rccl_path():
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return ""


def _agree(ok, group, dev):
    """True only if ``ok`` on EVERY rank of the group (one MIN all-reduce that every rank executes)."""
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return flag.item() == 1.0


def _create(group):
    """Communicator set-up in phases, each closed by an agreement collective that every rank executes, so that a failure on
    one rank (dlopen / dlsym, unique id, init, self-test) can never leave the others inside a different collective:
      1. bind RCCL (all ranks) and draw the unique id (rank 0)      -> agree
      2. broadcast the id (every rank, unconditionally after 1)
      3. ncclCommInitRank                                          -> agree
      4. self-test on the compute stream                           -> agree
    Returns the communicator or None (same answer on every rank)."""
    lib = hip.load()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    ident = (C.c_ubyte * 128)()
    ok, why = True, ""
    try:
        hip._check(lib.ucd_comm_load(_loaded_rccl_path().encode()), "ucd_comm_load")
        if rank == 0:
            hip._check(lib.ucd_comm_unique_id(C.addressof(ident), 128), "ucd_comm_unique_id")
    except Exception as e:                                       # binding problems: fall back, loudly
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        warnings.warn("direct RCCL communicator unavailable, using torch.distributed collectives: "
                      + (why or "another rank could not bind RCCL"))
        return None
    t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(t, src=src, group=group)
    ident = (C.c_ubyte * 128)(*t.cpu().tolist())
    handle = C.c_void_p()
    try:
        hip._check(lib.ucd_comm_init(C.addressof(ident), 128, world, rank, C.byref(handle)), "ucd_comm_init")
    except Exception as e:
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        warnings.warn("direct RCCL communicator unavailable, using torch.distributed collectives: "
                      + (why or "another rank failed ncclCommInitRank"))
        if ok and handle.value:
            lib.ucd_comm_destroy(handle.value)
        return None
    comm = DirectComm(handle.value, world, rank)
    # self-test on the compute stream: gather of the rank ids, sum of ones
    try:
        send = torch.full((4,), float(rank), device=dev)
        recv = torch.full((4 * world,), -1.0, device=dev)
        ones = torch.ones(4, device=dev)
        hip._check(lib.ucd_comm_all_gather(comm.handle, hip.ptr(send), hip.ptr(recv), 4, hip.stream()), "ucd_comm_all_gather")
        hip._check(lib.ucd_comm_all_reduce_sum(comm.handle, hip.ptr(ones), 4, hip.stream()), "ucd_comm_all_reduce_sum")
        expect = torch.arange(world, device=dev, dtype=torch.float32).repeat_interleave(4)
        ok = bool(torch.equal(recv, expect)) and bool(torch.equal(ones, torch.full((4,), float(world), device=dev)))
        why = "" if ok else "self-test mismatch"
    except Exception as e:
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        warnings.warn("direct RCCL communicator disabled: " + (why or "another rank failed its self-test"))
        return None
    return comm


def direct_comm(group=None):
    """``DirectComm`` for ``group`` (default group when None), or None when the direct path is unavailable."""
    key = id(group) if group is not None else None
    if key in _comms:
        return _comms[key]
    comm = None
    usable = (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()
              and dist.get_backend(group) == "nccl" and _switches.get("UCD_DIRECT_RCCL", "1") != "0")
    if usable:
        comm = _create(group)
    _comms[key] = comm
    return comm
