"""GPU probe (not part of the product): time ResNet-101/DeepLab-V3 forward+backward with a stock
BatchNorm+LeakyReLU stand-in under {fp32, bf16 autocast} x {NCHW, NHWC} to choose the activation
layout the hand-written kernels are designed around.  Usage: python tools/conv_layout_probe.py [B] [S]"""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from functools import partial
from ucd_amd.backbone import net_resnet101
from ucd_amd.blocks import DeeplabV3

class ShimABN(nn.BatchNorm2d):
    def __init__(self, c, activation="leaky_relu", activation_param=0.01):
        super().__init__(c); self.activation = activation; self.activation_param = activation_param
    def forward(self, x):
        y = super().forward(x)
        return F.leaky_relu(y, self.activation_param) if self.activation == "leaky_relu" else y

class Net(nn.Module):
    def __init__(self):
        super().__init__()
        na = partial(ShimABN, activation="leaky_relu", activation_param=0.01)
        self.body = net_resnet101(norm_act=na, output_stride=16)
        self.head = DeeplabV3(2048, 256, 256, norm_act=na, out_stride=16, pooling_size=32)
        self.cls = nn.Conv2d(256, 21, 1)
    def forward(self, x):
        return self.cls(self.head(self.body(x)))

def run(B, S, dtype, cl, steps=3):
    torch.manual_seed(0)
    net = Net().cuda().train()
    x = torch.randn(B, 3, S, S, device="cuda")
    if cl:
        net = net.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == "bf16")):
            y = net(x)
        y.float().mean().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    return dt

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 513
    print(torch.__version__, torch.cuda.get_device_name(0), "cpus", os.cpu_count())
    torch.backends.cudnn.benchmark = True
    for dtype in ("bf16", "fp32"):
        for cl in (True, False):
            try:
                dt = run(B, S, dtype, cl)
                print(json.dumps({"B": B, "S": S, "dtype": dtype, "channels_last": cl, "ms_fwd_bwd": dt * 1e3,
                                  "img_s": B / dt}), flush=True)
            except Exception as e:
                print("FAIL", dtype, cl, repr(e)[:300], flush=True)
    # raw HBM copy bandwidth
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print("copy GB/s (R+W):", 2 * a.numel() / dt / 1e9)
