"""GPU diagnostic: d bn3.bias of one bottleneck = column sums of dy (slope 1) - product (bf16 chain) and oracle (fp32 functional block)
against the sums taken directly.  usage: python tests/diag/bn3_bias_diag.py"""
import os, sys
from functools import partial
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ucd_amd import abn, blocks, synth, switches
from ucd_amd.ddp import DistributedDataParallel
from oracle import model as OM
DEV = "cuda:0"
for sa in ("1", "0"):
    switches.set("UCD_STAT_ATOMIC", sa)
    for cin, chans, hw in ((1024, (256, 256, 1024), 33), (256, (64, 64, 256), 129)):
        slope = 1.0
        norm = partial(abn.InPlaceABNSync, activation="leaky_relu", activation_param=slope)
        x0 = synth.t_normal(9, (24, cin, hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        dy = synth.t_normal(10, (24, chans[2], hw, hw), stream=1).to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        direct = dy.double().sum((0, 2, 3))
        blk = blocks.ResidualBlock(cin, chans, norm_act=norm, stride=1, dilation=1)
        state = synth.fill_state_dict(blk.state_dict(), 5)
        blk.load_state_dict(state)
        blk = blk.to(DEV).to(memory_format=torch.channels_last).train()
        mod = DistributedDataParallel(blk, bf16_weights=True)
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(x * 1.0)
        y.backward(dy)
        mod.finish_grad_sync()
        gb = blk.convs.bn3.bias.grad.double()
        P = {"blk." + k: v.to(DEV).float() for k, v in state.items()}
        for k, v in P.items():
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                v.requires_grad_(True)
        xo = x0.float().contiguous().clone().requires_grad_(True)
        yo = OM.residual_block(xo * 1.0, P, "blk", 1, 1, True, slope=slope)
        yo.backward(dy.float().contiguous())
        go = P["blk.convs.bn3.bias"].grad.double()
        rel = lambda a, b: ((a - b).norm() / b.norm()).item()
        print(f"UCD_STAT_ATOMIC={sa} {cin}->{chans} hw {hw}: product vs direct {rel(gb, direct):.2e}  oracle vs direct {rel(go, direct):.2e}  "
              f"product vs oracle {rel(gb, go):.2e}   |direct| rms {direct.pow(2).mean().sqrt().item():.1f}")
        # the same for d bn3.weight against a float64 evaluation from the oracle's own z3
        del blk, mod, x, y, P, xo, yo
        torch.cuda.empty_cache()
