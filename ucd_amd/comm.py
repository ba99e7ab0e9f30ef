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
        self.ipc = False          # the small collectives run on the one-shot IPC mailbox exchange
        self.ipc_tried = False    # the mailbox set-up ran once (whatever its outcome)
        self.has_rccl = False     # an RCCL communicator sits behind the handle (False: mailbox only)


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


IPC_SLOT_FLOATS = 32768      # per rank and exchange: the largest replicated statistics accumulator of a SyncBN layer (R x 2 C floats)


def _self_test(comm, group, dev):
    """The mailbox against the process group itself before anything depends on it: all-gather + all-reduce of seeded random vectors at
    the SyncBN message sizes (one chunk ... the whole slot, an odd length for the 4-byte path), several rounds each (both mailbox
    parities, back-to-back exchanges in flight), results required BIT-identical to torch.distributed's all_gather of the same vectors
    summed in rank order.  A node whose peer mappings misbehave (stale reads, lost stores) fails here, on every rank, and keeps RCCL.
    A rank whose library call raises keeps issuing the SAME c10d collectives as its peers: the verdict is then reached
    by the caller's agreement round, never by ranks sitting in different collectives."""
    lib = hip.load()
    world, rank = comm.world, comm.rank
    gen = torch.Generator(dev).manual_seed(977 + rank)
    ok, why = True, ""
    for n in (8, 13, 512, 4096, 16384, IPC_SLOT_FLOATS):
        sent, got = [], []
        for rnd in range(4):
            mine = torch.randn(n, device=dev, generator=gen)
            red = mine.clone()
            gat = torch.empty(world * n, device=dev)
            if ok:
                try:
                    hip._check(lib.ucd_comm_all_gather(comm.handle, hip.ptr(mine), hip.ptr(gat), n, hip.stream()), "ucd_comm_all_gather")
                    hip._check(lib.ucd_comm_all_reduce_sum(comm.handle, hip.ptr(red), n, hip.stream()), "ucd_comm_all_reduce_sum")
                except Exception as e:                            # no further library calls; the c10d side below goes on
                    ok, why = False, repr(e)
            sent.append(mine); got.append((gat, red))
        for mine, (gat, red) in zip(sent, got):
            ref = [torch.empty(n, device=dev) for _ in range(world)]
            dist.all_gather(ref, mine, group=group)
            acc = torch.zeros(n, device=dev)
            for r in ref:
                acc += r
            if ok and not (torch.equal(gat, torch.cat(ref)) and torch.equal(red, acc)):
                ok, why = False, "self-test mismatch at %d floats" % n
    if ok and lib.ucd_comm_ipc_timeouts(comm.handle):
        ok, why = False, "self-test timed out"
    return ok, why


def _attach_ipc(comm, group, dev):
    """Give ``comm`` a mailbox for the small collectives (csrc/comm.hip: one-shot IPC exchange; ``UCD_IPC_SYNC=0`` keeps RCCL).
    Phased like _create: every phase ends in an agreement collective, a failure anywhere drops the mailbox on EVERY rank.
    Returns True when the mailbox is in use."""
    lib = hip.load()
    world = comm.world
    nb = lib.ucd_comm_ipc_handle_bytes()
    mine = (C.c_ubyte * nb)()
    ok, why = True, ""
    try:
        hip._check(lib.ucd_comm_ipc_create(comm.handle, IPC_SLOT_FLOATS, int(_switches.get("UCD_IPC_TIMEOUT_MS", "60000")),
                                           C.addressof(mine)), "ucd_comm_ipc_create")
    except Exception as e:
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        lib.ucd_comm_ipc_drop(comm.handle)
        warnings.warn("IPC mailbox unavailable, the small collectives stay on RCCL: " + (why or "another rank could not create its mailbox"))
        return False
    gathered = [None] * world
    dist.all_gather_object(gathered, bytes(mine), group=group)
    table = (C.c_ubyte * (nb * world))(*b"".join(gathered))
    try:
        hip._check(lib.ucd_comm_ipc_connect(comm.handle, C.addressof(table)), "ucd_comm_ipc_connect")
    except Exception as e:
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        lib.ucd_comm_ipc_drop(comm.handle)
        warnings.warn("IPC mailbox unavailable, the small collectives stay on RCCL: " + (why or "another rank could not open a peer's mailbox"))
        return False
    try:
        ok, why = _self_test(comm, group, dev)
    except Exception as e:
        ok, why = False, repr(e)
    if not _agree(ok, group, dev):
        torch.cuda.synchronize()
        lib.ucd_comm_ipc_drop(comm.handle)
        warnings.warn("IPC mailbox disabled: " + (why or "another rank failed its self-test"))
        return False
    return True


def _create_local(group):
    """A communicator WITHOUT RCCL (several ranks on one GPU - RCCL refuses that -, or a non-nccl process group): the mailbox
    exchange alone.  None when the mailbox cannot be set up (same answer on every rank)."""
    lib = hip.load()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    handle = C.c_void_p()
    hip._check(lib.ucd_comm_init_local(world, rank, C.byref(handle)), "ucd_comm_init_local")
    comm = DirectComm(handle.value, world, rank)
    comm.ipc_tried = True
    comm.ipc = _attach_ipc(comm, group, dev)
    if not comm.ipc:
        lib.ucd_comm_destroy(comm.handle)
        return None
    return comm


def _ranks_own_their_devices(group):
    """True when no two ranks of the group drive the same GPU (collective).  The mailbox exchange spins on the device: ranks that
    SHARE a GPU are time-sliced by the hardware scheduler and every exchange then costs a scheduling quantum (measured: 10 ms at four
    processes on one MI355X, profiles/r05_ipc_exchange.md) - correct, and useless."""
    dev = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(dev)
    mine = (os.uname().nodename, str(getattr(props, "uuid", "")) or str(dev), getattr(props, "pci_bus_id", dev))
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    return len(set(everyone)) == len(everyone)


def _want_mailbox(group, explicit=False):
    """UCD_IPC_SYNC: "0" - never; "1" - always, as soon as the SyncBN communicator exists (also ranks that share a GPU: the tests);
    "auto" (default, round 6) - only where the caller asks for it EXPLICITLY (``attach_mailbox``: bench.py's third phase, after the
    RCCL phases have been measured) and every rank owns its GPU.  A plain ``run.py`` job therefore keeps its SyncBN exchanges on RCCL:
    the mailbox has never crossed xGMI on a box available to this build, and a training run must not be the first to try."""
    sw = _switches.get("UCD_IPC_SYNC", "auto")
    if sw == "0" or dist.get_world_size(group) < 2:
        return False
    if sw == "1":
        return True
    return bool(explicit) and _ranks_own_their_devices(group)


def _key(group, purpose="sync"):
    return (id(group) if group is not None else None, purpose)


def attach_mailbox(group=None):
    """Give the group's EXISTING SyncBN communicator its mailbox now (collective; bench.py does this after the RCCL-only phases have
    been measured).  True when the small collectives run on the mailbox from here on - graphs captured before still replay RCCL."""
    key = _key(group)
    comm = _comms.get(key)
    if comm is None:
        # no communicator yet (or a non-nccl group, which has no RCCL one): a mailbox-only communicator
        if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()) or dist.get_backend(group) == "nccl":
            return False
        if not _want_mailbox(group, explicit=True):
            return False
        comm = _comms[key] = _create_local(group)
        return comm is not None
    if comm.world < 2:
        return False
    if not comm.ipc:
        comm.ipc_tried = True
        comm.ipc = _want_mailbox(group, explicit=True) and _attach_ipc(comm, group, torch.device("cuda", torch.cuda.current_device()))
    return comm.ipc


def mailbox_timeouts(group=None):
    """Exchanges of the group's communicator that gave up waiting for a peer (0: none; their results are NaN and the communicator
    is poisoned on every rank - csrc/comm.hip)."""
    comm = _comms.get(_key(group))
    return int(hip.load().ucd_comm_ipc_timeouts(comm.handle)) if comm is not None and comm.ipc else 0


def check_mailbox(group=None):
    """Raise when an exchange of the group's mailbox timed out.  Trainer.train calls this at its host synchronisations: a replayed
    step graph issues no host-side collective call that could report the latched word."""
    n = mailbox_timeouts(group)
    if n:
        raise RuntimeError("SyncBN mailbox exchange %d timed out (a rank never wrote its vector): the statistics of every later "
                           "layer are NaN on every rank.  Run with UCD_IPC_SYNC=0 (RCCL exchanges)." % n)


def direct_comm(group=None, ipc=True, purpose="sync"):
    """``DirectComm`` for ``group`` (default group when None), or None when the direct path is unavailable.
    ``purpose``: "sync" - the SyncBN exchanges (compute stream; may carry the one-shot mailbox for small messages); "grad" - the
    gradient buckets of ucd_amd.ddp (the reducer's side stream).  The two are DIFFERENT communicators (round 6): the
    mailbox protocol keeps its sequence counters, slots and flags per communicator and is only safe for stream-ordered callers - a
    small gradient bucket routed to the SyncBN communicator's mailbox from the side stream could take the same sequence number as a
    SyncBN exchange in flight on the compute stream.  The "grad" communicator never gets a mailbox: its collectives are always RCCL."""
    if purpose != "sync":
        ipc = False
    key = _key(group, purpose)
    if key in _comms:
        comm = _comms[key]
        # UCD_IPC_SYNC=1: created by a caller that did not want the mailbox, asked for now (SyncBN): attach it once - collective over
        # the group like the creation itself, every rank reaches it at the same first SyncBN layer
        if comm is not None and ipc and not comm.ipc and not comm.ipc_tried and comm.world > 1:
            comm.ipc_tried = True
            comm.ipc = _want_mailbox(group) and _attach_ipc(comm, group, torch.device("cuda", torch.cuda.current_device()))
        return comm
    live = dist.is_available() and dist.is_initialized() and torch.cuda.is_available() and _switches.get("UCD_DIRECT_RCCL", "1") != "0"
    if not live:
        return None
    comm = None
    if dist.get_backend(group) == "nccl":
        comm = _create(group)
        if comm is not None:
            comm.has_rccl = True
            if ipc and comm.world > 1:
                comm.ipc_tried = True
                comm.ipc = _want_mailbox(group) and _attach_ipc(comm, group, torch.device("cuda", torch.cuda.current_device()))
    elif not ipc:
        return None                                   # a mailbox-only communicator is of no use to this caller; leave the slot open
    elif _want_mailbox(group):
        comm = _create_local(group)
    _comms[key] = comm
    return comm
