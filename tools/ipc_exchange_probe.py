"""Latency of the one-shot IPC mailbox exchange (csrc/comm.hip) between real processes sharing ONE GPU (the only multi-process
configuration a 1-GPU box offers; across GPUs each peer store / flag poll adds an xGMI hop).  Launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=W --master-addr 127.0.0.1 tools/ipc_exchange_probe.py [out.json]
Each rank captures a graph of 200 back-to-back all-reduces of n floats and replays it; rank 0 prints us per exchange for the SyncBN
message sizes (2 C floats ... R x 2 C floats)."""
import json, os, sys, time
os.environ.setdefault("UCD_IPC_SYNC", "1")     # the ranks share the GPU: "auto" would decline
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from ucd_amd import hip
from ucd_amd.comm import direct_comm

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
comm = direct_comm(None)
assert comm is not None and comm.ipc
lib = hip.load()
res = {"world": world, "ranks_on_one_gpu": True, "us_per_exchange": {}}
CHAIN = 200
s = torch.cuda.Stream()      # ONE side stream for all sizes: every further HIP queue of a process sharing the GPU is another
                             # candidate for the hardware scheduler's time slicing (10 ms quanta once the queues oversubscribe)
for n in (128, 512, 2048, 4096, 16384, 32768):
    buf = torch.zeros(n, device=dev)
    torch.cuda.synchronize(); dist.barrier()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(CHAIN):
                hip._check(lib.ucd_comm_all_reduce_sum(comm.handle, hip.ptr(buf), n, hip.stream()), "all_reduce")
        g.replay(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (5 * CHAIN) * 1e6
    res["us_per_exchange"][str(n)] = round(dt, 2)
    dist.barrier()
assert lib.ucd_comm_ipc_timeouts(comm.handle) == 0
if rank == 0:
    print(json.dumps(res))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"))
dist.destroy_process_group()
