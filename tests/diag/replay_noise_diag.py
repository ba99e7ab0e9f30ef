"""How far apart are two runs of the SAME mode of tests/test_step_gpu.py::_scheduled_steps, and does the order of the runs matter?
(The replay tests compare an eager run with a captured one built right after it in the same process.)
usage: python tests/diag/replay_noise_diag.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ucd_amd import switches
import test_step_gpu as T

switches.set("UCD_STAT_ATOMIC", "0")
runs = []
for mode in ("0", "0", "0", "1", "1"):
    losses, params, n_graph, _, err = T._scheduled_steps(mode)
    runs.append((mode, losses, params))
    print("mode", mode, "replayed", n_graph, "err", err, "step-2 losses", losses[1], flush=True)
ref = runs[0][1]
for i, (mode, l, _) in enumerate(runs[1:], 1):
    print(f"run {i} (UCD_STEP_GRAPH={mode}) vs run 0: max rel per step", np.round(np.abs(l - ref).max(1) / np.abs(ref).max(1), 6), flush=True)
l1, l2 = runs[1][1], runs[2][1]
print("run 2 vs run 1 (both eager, both not the first of the process):", np.round(np.abs(l2 - l1).max(1) / np.abs(l1).max(1), 6))
l3, l4 = runs[3][1], runs[4][1]
print("run 4 vs run 3 (both captured):", np.round(np.abs(l4 - l3).max(1) / np.abs(l3).max(1), 6))
print("run 3 (captured) vs run 2 (eager, not first):", np.round(np.abs(l3 - l2).max(1) / np.abs(l2).max(1), 6))
