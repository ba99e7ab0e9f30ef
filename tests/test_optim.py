"""The one-launch optimiser step (ucd_amd/optim.py, csrc/sgd.hip) - reference: ``optim.step()`` of
``torch.optim.SGD(momentum=0.9, nesterov=True)`` with per-group weight decay (train.py:147, run.py:175-186).

CPU: the host logic (block table, structure sizes, "no CPU path").  GPU: the kernel through the C ABI against the
update rule evaluated in float64 by torch with roundings at the same places (bit-exact), against ``torch.optim.SGD``
itself (a few ulp: its kernels round intermediates differently), and the bf16 working copies of ``ucd_amd.master``."""
import copy
import ctypes as C
import warnings

import numpy as np
import pytest
import torch

from oracle.optim import poly_lr, sgd_step as _reference_step
from ucd_amd import hip, optim


def test_block_table_covers_every_element_once():
    chunk = 4096
    sizes = [1, 4096, 4097, 10000, 64, 3 * 4096]
    tab = optim.block_table(sizes, chunk)
    assert tab.dtype == np.int32 and tab.shape[1] == 2
    seen = [np.zeros(n, dtype=np.int32) for n in sizes]
    for t, c in tab:
        seen[t][c * chunk:min(sizes[t], (c + 1) * chunk)] += 1
        assert c * chunk < sizes[t]
    assert all((s == 1).all() for s in seen)


def test_structures_match_the_header_and_empty_call_is_a_noop():
    # include/ucd_hip.h: ucd_sgd_tensor = 4 pointers + long long + 2 ints; ucd_sgd_hyper = 3 x 8 doubles + 8 ints
    assert C.sizeof(optim._SgdTensor) == 48 and C.sizeof(optim._SgdHyper) == 3 * 8 * 8 + 8 * 4
    lib = hip.load()
    assert lib.ucd_sgd_chunk() == 4096
    assert lib.ucd_sgd_step(None, None, 0, C.byref(optim._SgdHyper()), None) == 0
    assert lib.ucd_sgd_step(None, None, -1, C.byref(optim._SgdHyper()), None) != 0


def test_no_cpu_path():
    p = torch.nn.Parameter(torch.randn(8))
    p.grad = torch.randn(8)
    opt = optim.SGD([p], lr=0.1, momentum=0.9, nesterov=True)
    assert isinstance(opt, torch.optim.SGD)
    with pytest.raises(RuntimeError, match="GPU only"):
        opt.step()
    with pytest.raises(NotImplementedError):
        optim.SGD([p], lr=0.1, maximize=True)


def test_float64_rule_is_torch_sgd():
    """The rule the kernel is held to bit-exactly (``oracle.optim.sgd_step``) IS torch.optim.SGD's update (the optimiser of
    run.py:175-186) up to fp32 rounding of intermediates - checked on the CPU over several steps, groups and a PolyLR-like lr."""
    torch.manual_seed(0)
    for mu, nesterov, wd in ((0.9, True, 1e-4), (0.9, False, 0.0), (0.0, False, 5e-4)):
        p = torch.randn(3000)
        m = torch.zeros(3000)
        q = torch.nn.Parameter(p.clone())
        opt = torch.optim.SGD([q], lr=0.05, momentum=mu, nesterov=nesterov, weight_decay=wd)
        for step in range(5):
            lr = poly_lr(0.05, step, 10)
            opt.param_groups[0]["lr"] = lr
            g = torch.randn(3000)
            q.grad = g.clone()
            opt.step()
            p, m = _reference_step(p, g, m, lr, mu, wd, nesterov)
            torch.testing.assert_close(p, q.detach(), rtol=2e-6, atol=2e-6)
            if mu != 0:
                torch.testing.assert_close(m, opt.state[q]["momentum_buffer"], rtol=2e-6, atol=2e-6)


def test_poly_lr_restatement_is_the_scheduler():
    from ucd_amd.scheduler import PolyLR
    q = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([q], lr=0.01)
    sched = PolyLR(opt, max_iters=30, power=0.9)
    for it in range(1, 30):
        opt.step()
        sched.step()
        assert opt.param_groups[0]["lr"] == pytest.approx(poly_lr(0.01, it, 30), rel=1e-12)


# ---- GPU -------------------------------------------------------------------------------------------------------------

def _params(dev, seed=0):
    """Assorted tensors: channels-last 4-D weights (1x1 and 3x3), vectors, odd sizes, and views at odd element offsets of
    one flat buffer (16-byte alignment broken: the kernel's scalar path)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [(64, 32, 3, 3), (128, 64, 1, 1), (256,), (1,), (3,), (4097,), (21, 256, 1, 1), (5, 7, 3, 3)]
    ps = []
    for s in shapes:
        t = torch.randn(s, generator=g).to(dev)
        if len(s) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(t)
    flat = torch.randn(4300, generator=g).to(dev)
    return ps, flat


def _odd_views(flat):
    return [flat[1:101], flat[101:104], flat[105:105 + 4099]]


@pytest.mark.gpu
def test_step_is_bit_exact_against_the_float64_rule_and_close_to_torch():
    dev = torch.device("cuda:0")
    base, base_flat = _params(dev)
    groups_hp = [dict(momentum=0.9, nesterov=True, weight_decay=1e-4), dict(momentum=0.9, nesterov=False, weight_decay=0.0),
                 dict(momentum=0.0, nesterov=False, weight_decay=5e-4)]
    split = [base[0:4], base[4:8]]

    def make(cls, **kw):
        ps = [[torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)) if t.dim() == 4 else torch.nn.Parameter(t.clone())
               for t in part] for part in split]
        ps.append([torch.nn.Parameter(v) for v in _odd_views(base_flat.clone())])     # Parameters that ARE the odd-offset views
        assert all(p.data_ptr() % 16 != 0 for p in ps[2])
        return ps, cls([dict(params=pp, **hp) for pp, hp in zip(ps, groups_hp)], lr=0.05, **kw)

    mine_p, mine = make(optim.SGD)
    ref_p, ref = make(torch.optim.SGD, foreach=False)
    state_p = [[t.detach().clone(memory_format=torch.preserve_format) for t in part] for part in mine_p]
    state_m = [[torch.zeros_like(t) for t in part] for part in mine_p]
    gen = torch.Generator(device="cpu").manual_seed(1)
    for step in range(4):
        lr = poly_lr(0.05, step, 10)
        for opt in (mine, ref):
            for grp in opt.param_groups:
                grp["lr"] = lr
        for gi in range(3):
            for k, p in enumerate(mine_p[gi]):
                skip = step == 1 and gi == 0 and k == 2        # a parameter without gradient in one step: left alone
                grad = torch.randn(p.shape, generator=gen).to(dev)
                if p.dim() == 4:
                    grad = grad.contiguous(memory_format=torch.channels_last)
                for q in (p, ref_p[gi][k]):
                    q.grad = None if skip else grad.clone(memory_format=torch.preserve_format)
                if not skip:
                    hp = groups_hp[gi]
                    state_p[gi][k], state_m[gi][k] = _reference_step(state_p[gi][k], grad, state_m[gi][k], lr, hp["momentum"],
                                                                     hp["weight_decay"], hp["nesterov"])
        with warnings.catch_warnings():
            warnings.simplefilter("error")                     # the kernel path, not the torch fall-through
            mine.step()
        ref.step()
        for gi in range(3):
            for k, p in enumerate(mine_p[gi]):
                assert torch.equal(p.detach(), state_p[gi][k]), (step, gi, k)
                if groups_hp[gi]["momentum"] != 0:
                    assert torch.equal(mine.state[p]["momentum_buffer"], state_m[gi][k]), (step, gi, k)
                else:
                    assert "momentum_buffer" not in mine.state[p] or mine.state[p]["momentum_buffer"] is None
                torch.testing.assert_close(p.detach(), ref_p[gi][k].detach(), rtol=2e-6, atol=3e-6)
    # state_dict has torch's layout and loads into torch's class (and back, with buffers of another memory format)
    sd = mine.state_dict()
    assert set(sd) == {"state", "param_groups"} and all("momentum_buffer" in v for k, v in sd["state"].items() if k < 8)
    ref.load_state_dict(copy.deepcopy(sd))                  # (load_state_dict adopts the tensors it is given: no sharing)
    sd2 = copy.deepcopy(ref.state_dict())
    for v in sd2["state"].values():
        if v.get("momentum_buffer") is not None and v["momentum_buffer"].dim() == 4:
            v["momentum_buffer"] = v["momentum_buffer"].contiguous()       # NCHW, like a checkpoint of the reference
    mine.load_state_dict(sd2)
    before = [p.detach().clone() for p in mine_p[0]]
    mine.step()
    ref.step()
    for k, p in enumerate(mine_p[0]):
        assert not torch.equal(p.detach(), before[k])
        torch.testing.assert_close(p.detach(), ref_p[0][k].detach(), rtol=2e-6, atol=3e-6)


@pytest.mark.gpu
def test_step_writes_the_bf16_working_copies():
    """ucd_amd.master: the optimiser's launch leaves bf16(master) in the working copies (bit-exact with ``.to(bfloat16)``),
    the cast kernel is skipped and only the flipped / transposed set is refreshed afterwards."""
    from ucd_amd.blocks import Conv1x1, Conv3x3
    from ucd_amd.master import Bf16Weights
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = torch.nn.Sequential(Conv3x3(64, 64, 3, padding=1, bias=False), Conv1x1(64, 128),
                              Conv3x3(128, 64, 3, padding=2, dilation=2, bias=False)).to(dev).to(memory_format=torch.channels_last)
    w16 = Bf16Weights(net)
    opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    for step in range(2):
        for m in net:
            m.weight.grad = torch.randn_like(m.weight)
        copies = [m._w16.detach().clone() for m in net]
        opt.step()
        assert w16._dirty is False and w16._flips_dirty is True
        for m, old in zip(net, copies):
            assert torch.equal(m._w16.detach(), m.weight.detach().to(torch.bfloat16))
            assert not torch.equal(m._w16.detach(), old)
        w16.refresh_if_stale()
        assert w16._flips_dirty is False
        for m in net:
            flip = getattr(m, "_w16_flip", None)
            if flip is not None:
                assert torch.equal(flip, m._w16.detach().flip(2, 3).transpose(0, 1))
    # another optimiser class stepping the same weights: the hook falls back to the full refresh
    other = torch.optim.SGD(net.parameters(), lr=0.1)
    other.step()
    assert w16._dirty is True
    w16.refresh_if_stale()
    for m in net:
        assert torch.equal(m._w16.detach(), m.weight.detach().to(torch.bfloat16))


@pytest.mark.gpu
def test_launcher_builds_the_hip_step_by_default(monkeypatch):
    from ucd_amd import argparser, tasks
    from ucd_amd.run import build_models, make_optimizer
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(
        ["--method", "UCD", "--task", "15-5", "--step", "0", "--no_pretrained"]))
    model = build_models(opts, torch.device("cuda:0"), tasks.get_per_task_classes("voc", "15-5", 0))[0]
    from ucd_amd import switches
    monkeypatch.delenv("UCD_SGD", raising=False)
    switches.reload()
    opt = make_optimizer(opts, model)
    assert isinstance(opt, optim.SGD) and len(opt.param_groups) == 3       # the one-launch step is what run.py:175-186 builds
    monkeypatch.setenv("UCD_SGD", "torch")
    switches.reload()
    try:
        assert type(make_optimizer(opts, model)) is torch.optim.SGD         # the A/B reference
    finally:
        monkeypatch.delenv("UCD_SGD", raising=False)
        switches.reload()
