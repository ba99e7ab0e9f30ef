"""Does a training iteration read memory it has not written?  Three iterations of tests/test_step_gpu.py::_scheduled_steps (eager,
deterministic statistics) on a fresh process, then every scratch buffer of the Python and C++ layers filled with a NaN pattern and
a few GiB of freed allocator blocks left full of NaNs, then the same three iterations from a freshly built model: the losses must be
bit-identical.  usage: python tests/diag/poison_step_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from ucd_amd import abn, hip, switches
import test_step_gpu as T

switches.set("UCD_STAT_ATOMIC", os.environ.get("DIAG_STAT_ATOMIC", "0"))
dev = torch.device("cuda:0")
a, _, _, _, _ = T._scheduled_steps("0", steps=3)
print("fresh    ", a[:, 3], flush=True)

WHAT = os.environ.get("DIAG_POISON", "py,cpp,alloc").split(",")
ONLY = os.environ.get("DIAG_POISON_KEY", "")

def poison(pattern, value):
    if "py" in WHAT:
        for k, buf in hip._workspaces.items():
            if ONLY and ONLY not in str(k):
                continue
            buf.view(torch.int32)[: buf.numel() * buf.element_size() // 4].fill_(pattern)
    node = abn._abn_node()
    if "cpp" in WHAT and node is not None and hasattr(node, "poison_workspaces"):
        node.poison_workspaces(int(os.environ.get("DIAG_POISON_BYTE", "255")), int(os.environ.get("DIAG_POISON_TAG", "-1")))                         # 0xFFFFFFFF: a NaN in fp32, in bf16 pairs too
    if "dummy" in WHAT:                                      # the same kind of work on memory nothing uses
        global _dummy
        _dummy = torch.empty(96 << 20, dtype=torch.uint8, device=dev)
        for _ in range(6):
            _dummy.fill_(0)
    if "alloc" in WHAT:
        blocks = [torch.full((64 << 20,), value, device=dev) for _ in range(12)]
        small = [torch.full((n,), value, device=dev) for n in (64, 256, 512, 2048, 8192, 65536, 1 << 20, 1 << 22) for _ in range(64)]
        del blocks, small
print("python workspaces:", [(str(k), int(v.numel() * v.element_size())) for k, v in hip._workspaces.items()], flush=True)

poison(0x7F800001, float("nan"))
b, _, _, _, _ = T._scheduled_steps("0", steps=3)
print("poisoned ", b[:, 3], "identical:", np.array_equal(a, b), "nan:", bool(np.isnan(b).any()), flush=True)
poison(0x7149F2CA, 1e30)
c, _, _, _, _ = T._scheduled_steps("0", steps=3)
print("huge     ", c[:, 3], "identical:", np.array_equal(a, c), flush=True)
if not (np.array_equal(a, b) and np.array_equal(a, c)):
    print("per-step max rel difference poisoned vs fresh:", np.abs(b - a).max(1) / np.abs(a).max(1), " huge vs fresh:", np.abs(c - a).max(1) / np.abs(a).max(1))
