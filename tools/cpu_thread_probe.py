"""Host probe: oracle teacher forward (2 x 513^2) at several thread counts, to size bench.py's cpu_baseline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model as OM, step as OS
from oracle.params import template_state
from ucd_amd import synth
P = OS.make_params(synth.fill_state_dict(template_state([16]), 42), requires_grad=False)
img = synth.images(1, 2, 513)
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        OM.segmentation_forward(img[:1, :, :129, :129], P, 1, training=False)
        t0 = time.time(); OM.segmentation_forward(img, P, 1, training=False); dt = time.time() - t0
    print(f"threads {nt}: teacher fwd 2x513^2 {dt:.2f}s", flush=True)
