"""Probe: collectives on CUDA tensors through gloo with two ranks sharing cuda:0 (multi-rank code paths on a 1-GPU box)."""
import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
for name, fn in [
    ("all_reduce", lambda: dist.all_reduce(torch.full((4,), float(r + 1), device="cuda:0"))),
    ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(8, device="cuda:0"), torch.full((4,), float(r), device="cuda:0"))),
    ("broadcast", lambda: dist.broadcast(torch.full((4,), float(r), device="cuda:0"), src=0)),
    ("reduce", lambda: dist.reduce(torch.ones(1, device="cuda:0"), dst=0)),
    ("async all_reduce", lambda: dist.all_reduce(torch.ones(4, device="cuda:0"), async_op=True).wait()),
]:
    try:
        fn(); torch.cuda.synchronize(); print("rank", r, name, "OK", flush=True)
    except Exception as e:
        print("rank", r, name, "FAILED", repr(e)[:150], flush=True)
dist.destroy_process_group()
