"""Do the weight-gradient products fill the holes of the input-gradient chain?  The backward GEMMs of one layer3 bottleneck at 24
images (M = 26136: dgrad / wgrad of conv3 1x1 1024<-256, conv2 3x3 256, conv1 1x1 256<-1024), x8 blocks, captured in a graph:
(a) all on one stream in backward order, (b) input gradients on the main stream, weight gradients on a side stream forked after
each layer's dZ exists and joined at the end.  usage: python tools/overlap_probe.py [images]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucd_amd import hip

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H = W = 33
M = B * H * W
bf = torch.bfloat16
def r(*s): return torch.randn(*s, device=dev).to(bf)
dy3, y2, w3t = r(M, 1024), r(M, 256), r(256, 1024) * 0.03        # conv3: dX2 = dY3 . W3 ; dW3 = dY3^T . Y2
dz2, y1, w2t = r(M, 256), r(M, 256), r(256, 9 * 256) * 0.02      # conv2 3x3
dz1, x0, w1t = r(M, 256), r(M, 1024), r(1024, 256) * 0.06        # conv1: dX0 = dZ1 . W1 ; dW1 = dZ1^T . X0
dx2, dx1, dx0 = torch.empty(M, 256, device=dev, dtype=bf), torch.empty(M, 256, device=dev, dtype=bf), torch.empty(M, 1024, device=dev, dtype=bf)
dw3, dw2, dw1 = (torch.empty(1024, 256, device=dev), torch.empty(256, 9 * 256, device=dev), torch.empty(256, 1024, device=dev))
side = torch.cuda.Stream()
BLOCKS = 8

def dgrads(i):
    if i == 3: hip.conv1x1(dy3, w3t, dx2)
    if i == 2: hip.conv1x1(dz2, w2t, dx1, conv3=(H, W, 1))
    if i == 1: hip.conv1x1(dz1, w1t, dx0)
def wgrads(i):
    if i == 3: hip.conv_wgrad(dy3, y2, dw32=dw3)
    if i == 2: hip.conv_wgrad(dz2, y1, dw32=dw2, conv3=(H, W, 1))
    if i == 1: hip.conv_wgrad(dz1, x0, dw32=dw1)

def serial():
    for _ in range(BLOCKS):
        for i in (3, 2, 1):
            dgrads(i); wgrads(i)
def only(which):
    def f():
        for _ in range(BLOCKS):
            for i in (3, 2, 1):
                which(i)
    return f
def forked():
    main = torch.cuda.current_stream()
    for _ in range(BLOCKS):
        for i in (3, 2, 1):
            ev = torch.cuda.Event(); ev.record(main)        # dZ of this layer exists
            side.wait_event(ev)
            with torch.cuda.stream(side):
                wgrads(i)
            dgrads(i)
    ev = torch.cuda.Event(); ev.record(side); main.wait_event(ev)

def graph_time(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): g.replay()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3 / BLOCKS

t_d, t_w = graph_time(only(dgrads)), graph_time(only(wgrads))
t_s, t_f = graph_time(serial), graph_time(forked)
print(f"images {B}: per bottleneck  dgrads alone {t_d:7.1f} us  wgrads alone {t_w:7.1f}  one stream {t_s:7.1f}  forked {t_f:7.1f}  "
      f"({100 * (1 - t_f / t_s):.1f} % less)", flush=True)
