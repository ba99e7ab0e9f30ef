"""Host-side profile of the train step at a small per-rank batch (the 8-GPU regime): where does Python time go?"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("UCD_ABN_FORCE_SYNC") == "1":      # the multi-rank code path (collectives over a 1-rank RCCL group)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1)
import bench
sys.argv = ["bench.py", "--global_batch", "3"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
torch.backends.cudnn.benchmark = True
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, 3, 0)
for _ in range(5): trainer.train_step(images, labels, optimizer, scheduler)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): trainer.train_step(images, labels, optimizer, scheduler)
t_host = time.perf_counter() - t0          # enqueue only: what the host needs per step when the GPU is not the limit
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 100, " host enqueue ms/step", t_host * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): trainer.train_step(images, labels, optimizer, scheduler)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000]); s2 = io.StringIO(); pstats.Stats(pr, stream=s2).sort_stats("cumulative").print_stats(45); print(s2.getvalue()[:9000])
