"""Round 6: do the ranks of the one-shot mailbox exchange (csrc/comm.hip) ever part?  W real processes share ONE
GPU (the configuration in which profiles/r05_ipc_exchange.md recorded a run out of lockstep and a 15-minute "hang").  Every rank issues
the same sequence of all-reduces of mixed SyncBN sizes with other kernels in between, keeps a checksum of every result, and every
CHECK exchanges the checksums are all-gathered over gloo: the first exchange whose result differs between ranks is reported, with
the number of timed-out exchanges in front of it.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 tools/ipc_lockstep_harness.py [exchanges] [timeout ms] [stall]
stall = "rank:k:seconds": that rank's HOST sleeps in front of exchange k (a rank that falls behind by more than the timeout - what a
2 s timeout met under four time-sliced processes in round 5).
Prints one line per run on rank 0: LOCKSTEP_RUN {"world":..,"exchanges":..,"timeouts":..,"first_diff":null|k,"s":..}"""
import json, os, sys, time
os.environ["UCD_IPC_SYNC"] = "1"
if len(sys.argv) > 2:
    os.environ["UCD_IPC_TIMEOUT_MS"] = sys.argv[2]
os.environ.setdefault("UCD_IPC_TIMEOUT_MS", "60000")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from ucd_amd import hip
from ucd_amd.comm import direct_comm

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
STALL = [float(x) for x in sys.argv[3].split(":")] if len(sys.argv) > 3 else None
CHECK = 50
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
comm = direct_comm(None)
assert comm is not None and comm.ipc
lib = hip.load()
sizes = (512, 2048, 4096, 1024, 16384, 13, 32768, 8192)
gen = torch.Generator(dev).manual_seed(4242 + rank)
filler = torch.randn(1 << (15 + rank % 3), device=dev)
sums = torch.zeros(N, device=dev, dtype=torch.float64)
first_diff, t0 = None, time.time()
for k in range(N):
    n = sizes[k % len(sizes)]
    buf = torch.randn(n, device=dev, generator=gen)
    if STALL is not None and rank == int(STALL[0]) and k == int(STALL[1]):
        torch.cuda.synchronize()
        time.sleep(STALL[2])
    try:
        hip._check(lib.ucd_comm_all_reduce_sum(comm.handle, hip.ptr(buf), n, hip.stream()), "all_reduce")
    except RuntimeError as e:                       # a latched timeout: keep going on NaN so that every rank reaches the checks
        buf.fill_(float("nan"))
    sums[k] = buf.double().sum() + buf[:8].double().sum() * 3.0
    if (k + rank) % 5 == 0:
        filler = filler * 1.0001 + 0.5              # ranks drift against each other
    if (k + 1) % CHECK == 0 or k + 1 == N:
        lo = (k // CHECK) * CHECK
        mine = sums[lo:k + 1].cpu()
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if first_diff is None:
            for j in range(mine.numel()):
                vals = [e[j].item() for e in every]
                if any((v != vals[0]) and not (v != v and vals[0] != vals[0]) for v in vals):
                    first_diff = lo + j
                    break
t = torch.tensor([float(lib.ucd_comm_ipc_timeouts(comm.handle) != 0)])
dist.all_reduce(t)
nan_any = bool(torch.isnan(sums).any().item())
if rank == 0:
    print("LOCKSTEP_RUN " + json.dumps({"world": world, "exchanges": N, "ranks_with_timeouts": int(t.item()), "first_diff": first_diff,
                                        "nan": nan_any, "timeout_ms": int(os.environ["UCD_IPC_TIMEOUT_MS"]), "s": round(time.time() - t0, 1)}), flush=True)
dist.barrier()
dist.destroy_process_group()
