"""Do the teacher's forward and the student's forward overlap in a graph replay?  At 3 images per GPU neither fills the chip, yet the
kernel trace of the captured step shows one behind the other (profiles/r06_side_stream.md).  Captures the two forwards (no autograd)
(1) serially in one graph, (2) forked / joined inside one graph - the shape of the captured step, (3) as TWO graphs replayed on two
streams, (4) each alone; prints ms per replay.   usage: python tools/teacher_overlap_probe.py [images per GPU]"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = types.SimpleNamespace(task="15-5", dataset="voc", step=1, crop=513, opt_level="O1", global_batch=B, pixcon_precision=None)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
trainer, optimizer, scheduler, images, labels, classes = bench.build(args, dev, B, 0)
model = trainer.model
up = {"upsample": False}

def teacher():
    return trainer._teacher_eager(images, up)
def student():
    with torch.no_grad(), trainer._autocast():
        return model(images, ret_intermediate=False, **up)

for _ in range(3):
    teacher(); student()
torch.cuda.synchronize()
main = torch.cuda.current_stream()
side = torch.cuda.Stream(dev)

def capture(fn, stream=None):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    if stream is None:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            keep = fn()
    else:
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            keep = fn()
    return g, keep

def serial():
    return teacher(), student()
def forked():
    m = torch.cuda.current_stream()
    side.wait_stream(m)
    with torch.cuda.stream(side):
        a = teacher()
    b = student()
    m.wait_stream(side)
    return a, b

def clock(run, n=30):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

g_serial, k1 = capture(serial)
g_fork, k2 = capture(forked)
g_t, k3 = capture(teacher)
g_s, k4 = capture(student)
side2 = torch.cuda.Stream(dev)
def two_graphs():
    m = torch.cuda.current_stream()
    side2.wait_stream(m)
    with torch.cuda.stream(side2):
        g_t.replay()
    g_s.replay()
    m.wait_stream(side2)
print("images per GPU", B)
print("teacher alone            %.3f ms" % clock(g_t.replay))
print("student forward alone    %.3f ms" % clock(g_s.replay))
print("one graph, serial        %.3f ms" % clock(g_serial.replay))
print("one graph, fork / join   %.3f ms" % clock(g_fork.replay))
print("two graphs, two streams  %.3f ms" % clock(two_graphs))
