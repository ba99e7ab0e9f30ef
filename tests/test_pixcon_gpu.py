"""GPU parity tests proper: the HIP contrastive path, called through the C ABI, against the pinned CPU
oracle on the same seeded inputs and against the golden vectors captured from the reference.
Bars: labels / counts / index sets bit-exact; float32 loss and gradients within 1e-3 relative
(north_star), in practice ~1e-5."""
import numpy as np
import pytest
import torch

from conftest import assert_matches_compact, load_golden
from oracle import contrastive as OC
from ucd_amd import synth

pytestmark = pytest.mark.gpu
PIXCON = ["voc_15_5", "city_13_6", "voc_15_5s_step2", "voc_19_1_odd"]


def _case(g):
    cfg = [int(v) for v in g["cfg"]]
    seed, B, N, h, w, K, H, W = cfg[:8]
    return synth.contrastive_case(seed, B, N, h, w, K, H, W, cfg[8:])


def _to_dev(*ts):
    dev = torch.device("cuda:0")
    return [t.to(dev) for t in ts]


@pytest.mark.parametrize("name", PIXCON)
def test_reference_shaped_api_matches_golden(name):
    """pre_contractive_pixel -> (a, c, la, lc, P) in the reference's order; PixelConLossV2 on it."""
    from ucd_amd.contrastive import PixelConLossV2, pre_contractive_pixel
    g = load_golden(f"pixcon_{name}.npz")
    f_n, f_o, l_po, labels = _case(g)
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    tup = pre_contractive_pixel(fn_d, lab_d, l_po=lpo_d, f_o=fo_d)
    a, c, la, lc, P = tup
    assert a.shape[0] == int(g["A"]) and c.shape[0] == int(g["C"])
    np.testing.assert_array_equal(la.cpu().numpy(), g["la"])           # bit-exact integer work
    np.testing.assert_array_equal(lc.cpu().numpy(), g["lc"])
    assert_matches_compact(g, "a", a.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert_matches_compact(g, "c", c.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert_matches_compact(g, "P", P.cpu().numpy(), rtol=1e-4, atol=1e-6)
    loss = PixelConLossV2(temperature=0.07)(tup)
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-3)
    assert abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])) < 1e-4
    loss.backward()
    assert_matches_compact(g, "grad_f_n", fn_d.grad.cpu().numpy(), rtol=1e-3, atol=1e-7)
    # P = None variant
    tup2 = pre_contractive_pixel(fn_d.detach(), lab_d, l_po=lpo_d, f_o=fo_d)
    l2 = PixelConLossV2(temperature=0.07)(tup2[0], tup2[1], tup2[2], tup2[3], None, batch=tup2.batch)
    assert l2.item() == pytest.approx(float(g["loss_noP"]), rel=1e-4)


@pytest.mark.parametrize("name", PIXCON)
def test_fused_loss_matches_oracle_and_golden(name):
    """The fused trainer path (rows grouped by label, positives-only second sweep)."""
    from ucd_amd.contrastive import ucd_contrastive_loss
    g = load_golden(f"pixcon_{name}.npz")
    f_n, f_o, l_po, labels = _case(g)
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    loss = ucd_contrastive_loss(fn_d, lab_d, lpo_d, fo_d, 0.07, 20)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])) < 1e-4
    assert_matches_compact(g, "grad_f_n", fn_d.grad.cpu().numpy(), rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("B,N,h,K,H,new_ids,max_label", [
    (3, 256, 33, 16, 513, list(range(16, 21)), 20),      # VOC 15-5 per-rank shape of the 8-GPU config
    (2, 256, 16, 14, 256, list(range(14, 20)), 20),      # Cityscapes-like
    (2, 64, 12, 101, 192, list(range(101, 151)), 150),   # ADE: labels beyond int8's 20-clamp (generalised bound)
    (2, 32, 9, 16, 129, [16], 20),
    (3, 256, 32, 101, 512, list(range(101, 151)), 150),  # BASELINE configs[3]: ADE 100-50 per-rank shape (3 x 512^2, K = 101)
    (2, 256, 48, 14, 768, list(range(14, 20)), 20),      # BASELINE configs[4]: Cityscapes 13-6 per-rank shape (2 x 768^2)
    (3, 256, 33, 18, 513, [18], 20),                     # BASELINE configs[2]: VOC 15-5s step 3 per-rank shape (K = 18)
])
def test_prep_and_loss_vs_oracle(B, N, h, K, H, new_ids, max_label):
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare, ucd_contrastive_loss
    f_n, f_o, l_po, labels = synth.contrastive_case(1000 + N + h, B, N, h, h, K, H, H, new_ids)
    ref_in = f_n.clone().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(ref_in, labels, l_po, f_o, max_label=max_label)
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    ref.backward()
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    # integer side, unsorted order == oracle order
    pb = pixcon_prepare(fn_d.detach(), lab_d, lpo_d, fo_d, max_label=max_label, sort_by_label=False)
    m = pb.meta_host()
    assert (m.A, m.Co, m.min_new) == (int(prep["keep"].sum()), int(prep["keep_o"].sum()), prep["min_new"])
    keep_idx = torch.nonzero(prep["keep"])[:, 0].int()
    old_idx = torch.nonzero(prep["keep_o"])[:, 0].int()
    assert torch.equal(pb.anchor_pix[:m.A].cpu(), keep_idx)
    assert torch.equal(pb.old_pix[:m.Co].cpu(), old_idx)
    assert torch.equal(pb.row_label[:m.A].cpu().long(), prep["la"])
    assert torch.equal(pb.row_label[m.Apad:m.Apad + m.Co].cpu().long(), prep["lc"][m.A:])
    assert m.n_valid == int(((prep["la"].view(-1, 1) == prep["lc"].view(1, -1)).sum(1) - 1 > 0).sum())
    torch.testing.assert_close(pb.pcat[:m.A, :K].cpu(), prep["pa"], rtol=1e-5, atol=1e-7)
    # sorted order: same sets, grouped by label, stable
    pbs = pixcon_prepare(fn_d.detach(), lab_d, lpo_d, fo_d, max_label=max_label, sort_by_label=True)
    ms = pbs.meta_host()
    order = torch.argsort(prep["la"], stable=True)
    assert torch.equal(pbs.anchor_pix[:ms.A].cpu(), keep_idx[order])
    assert torch.equal(pbs.row_label[:ms.A].cpu().long(), prep["la"][order])
    # row statistics and loss, both orders
    for batch in (pb, pbs):
        loss_out, grad_a, stats = pixcon_loss_raw(batch, 0.07, True, True, need_grad=True, row_stats=True)
        assert abs(loss_out[0].item() - ref.item()) / abs(ref.item()) < 1e-4
        assert int(loss_out[1].item()) == m.n_valid
    _, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True)
    torch.testing.assert_close(stats[0, :m.A].cpu().double(), neg, rtol=1e-4, atol=0)
    torch.testing.assert_close(stats[1, :m.A].cpu().double(), num, rtol=0, atol=0)
    torch.testing.assert_close(grad_a[:m.A, :N].cpu().double(), da, rtol=1e-3, atol=1e-4 * da.abs().max().item())
    # end to end through autograd
    loss = ucd_contrastive_loss(fn_d, lab_d, lpo_d, fo_d, 0.07, max_label)
    loss.backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-4
    scale = ref_in.grad.abs().max().item()
    assert (fn_d.grad.cpu() - ref_in.grad).abs().max().item() / scale < 1e-3


def test_label_downsample_bit_exact_full_size():
    """513x513 -> 33x33 at batch 24 (BASELINE config 2): every down-sampled label equals torch's."""
    from ucd_amd.contrastive import pixcon_prepare
    B, H, h, K = 24, 513, 33, 16
    labels = synth.seg_labels(77, B, H, H, range(16, 21))
    ds = OC.downsample_labels(labels, h, h, 20)
    dev = torch.device("cuda:0")
    f = torch.zeros(B, 32, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    lpo = torch.full((B, K, h, h), -5.0, device=dev)
    lpo[:, 0] = 5.0                                   # teacher says background everywhere
    pb = pixcon_prepare(f, labels.to(dev), lpo, f, max_label=20, sort_by_label=False)
    m = pb.meta_host()
    flat = ds.reshape(-1)
    assert m.A == int((flat > 0).sum()) and m.Co == 0 and m.min_new == int(flat[flat > 0].min())
    assert torch.equal(pb.anchor_pix[:m.A].cpu().long(), torch.nonzero(flat > 0)[:, 0])
    assert torch.equal(pb.row_label[:m.A].cpu().long(), flat[flat > 0])


def test_bf16_inputs_and_no_new_pixels():
    from ucd_amd.contrastive import ucd_contrastive_loss
    f_n, f_o, l_po, labels = synth.contrastive_case(5, 2, 256, 9, 9, 16, 129, 129, range(16, 21))
    fb, fob = f_n.bfloat16(), f_o.bfloat16()
    ref_in = fb.float().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(ref_in, labels, l_po, fob.float())
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    ref.backward()
    dev = torch.device("cuda:0")
    x = fb.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    loss = ucd_contrastive_loss(x, labels.to(dev), l_po.to(dev), fob.to(dev), 0.07, 20)
    loss.backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-4      # same bf16-rounded inputs, fp32 math
    assert x.grad.dtype == torch.bfloat16
    scale = ref_in.grad.abs().max().item()
    assert (x.grad.float().cpu() - ref_in.grad).abs().max().item() / scale < 1e-2


@pytest.mark.parametrize("B,N,h,K,H,new_ids,max_label", [
    (3, 256, 33, 16, 513, list(range(16, 21)), 20),
    (2, 64, 12, 101, 192, list(range(101, 151)), 150),
    (2, 32, 9, 16, 129, [16], 20),
    (2, 128, 16, 20, 256, [20], 20),          # 20 teacher classes: two 16-class steps of the probability products
])
def test_fp16_performance_mode_vs_oracle(B, N, h, K, H, new_ids, max_label):
    """fp16-operand sweep (v_mfma_f32_32x32x16_f16, online negative-max rescale): loss within 1e-3 relative
    of the fp32 oracle (north_star bar); gradient within 2e-3 of its largest entry (stated tolerance of the
    performance mode: operand rounding 2^-11 on unit vectors)."""
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare, ucd_contrastive_loss
    f_n, f_o, l_po, labels = synth.contrastive_case(2000 + N + h, B, N, h, h, K, H, H, new_ids)
    ref_in = f_n.clone().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(ref_in, labels, l_po, f_o, max_label=max_label)
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    ref.backward()
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    _, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    for sort in (False, True):
        pb = pixcon_prepare(fn_d.detach(), lab_d, lpo_d, fo_d, max_label=max_label, sort_by_label=sort, fp16=True)
        loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True, precision="f16")
        assert abs(loss_out[0].item() - ref.item()) / abs(ref.item()) < 1e-3
        if not sort:
            m = pb.meta_host()
            torch.testing.assert_close(stats[0, :m.A].cpu().double(), neg, rtol=2e-3, atol=0)
            err = (grad_a[:m.A, :N].cpu().double() - da).abs().max().item() / da.abs().max().item()
            assert err < 2e-3, err
    loss = ucd_contrastive_loss(fn_d, lab_d, lpo_d, fo_d, 0.07, max_label, "f16")
    loss.backward()
    assert abs(loss.item() - ref.item()) / abs(ref.item()) < 1e-3
    scale = ref_in.grad.abs().max().item()
    assert (fn_d.grad.cpu() - ref_in.grad).abs().max().item() / scale < 2e-3


@pytest.mark.parametrize("prec", ["f32", "f16", "f16_split"])
@pytest.mark.parametrize("case", ["no_anchor", "one_label", "single_anchor_block", "constant_features"])
def test_degenerate_batches_every_precision(prec, case):
    """Edge cases of the anchor / contrast sets through every loss path (the planned fp16 sweeps build their work lists on
    the device from these counts): no anchor at all (all-background labels and a teacher that predicts background: zero
    units, loss 0, zero gradient); every pixel carrying the same label (sweep 1 has nothing but the boundary / padding
    tiles, every row is a positive of every anchor); fewer anchors than one 128-row block; a constant student feature map
    (every anchor row identical: each positive ties with the self pair for the row maximum, which the planned sweeps seed
    with S_ii instead of visiting the positives)."""
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare
    B, N, h, K, H = 2, 256, 12, 16, 192
    f_n, f_o, l_po, labels = synth.contrastive_case(31, B, N, h, h, K, H, H, list(range(16, 21)))
    if case == "no_anchor":
        labels = torch.zeros_like(labels)
        l_po = l_po.clone(); l_po[:, 0] += 50.0                     # the teacher says background everywhere
    elif case == "one_label":
        labels = torch.full_like(labels, 17)
    elif case == "constant_features":
        f_n = f_n[:1, :, :1, :1].expand_as(f_n).contiguous()
    else:
        labels = torch.zeros_like(labels); labels[0, :40, :40] = 18
        l_po = l_po.clone(); l_po[:, 0] += 50.0
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    pb = pixcon_prepare(fn_d.contiguous(memory_format=torch.channels_last), lab_d, lpo_d, fo_d, sort_by_label=True, fp16=prec != "f32")
    loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True, precision=prec)
    m = pb.meta_host()
    assert torch.isfinite(loss_out).all() and torch.isfinite(grad_a[:max(m.A, 1)]).all()
    if case == "no_anchor":
        # (the reference itself stops here: min() of an empty tensor, utils/utils.py:353; the kernels define the loss as 0)
        assert m.A == 0 and loss_out[0].item() == 0.0 and loss_out[1].item() == 0.0
        return
    prep = OC.pre_contrastive_pixel(f_n, labels, l_po, f_o)
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    _, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    assert m.A == prep["a"].shape[0] and (case != "single_anchor_block" or m.A < 128)
    tol = 1e-4 if prec == "f32" else 1e-3
    assert abs(loss_out[0].item() - ref.item()) <= tol * abs(ref.item()) + 1e-5        # one_label: the loss is 0 up to rounding
    if case == "one_label":
        assert stats[0, :m.A].abs().max().item() == 0.0              # no negatives at all
    scale = max(da.abs().max().item(), 1e-12)
    # label-sorted rows -> the oracle's pixel order
    got = grad_a[:m.A, :N].cpu().double()[torch.argsort(pb.anchor_pix[:m.A].cpu())]
    assert (got - da).abs().max().item() / scale < (1e-4 if prec == "f32" else 3e-3)


@pytest.mark.parametrize("T", [0.07, 0.05])
def test_fp16_extreme_logit_range(T):
    """The widest logit range the fp16 sweep can meet: anchors whose first contrast tiles are far (cos ~ -1) and whose
    later tiles are near (cos ~ +1), i.e. 2/T*log2(e) = 41 (T = 0.07) / 58 (T = 0.05) log2 units between them.
    T = 0.07 takes the constant-shift form of sweep 1 (far terms underflow, as they may: they are < 2^-28 of the near
    ones); T = 0.05 is past its limit and takes the online-maximum form, whose rescale (threshold 2^8) fires at a
    chosen tile."""
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare
    B, N, h, K, H = 2, 64, 16, 16, 256
    f_n, f_o, l_po, labels = synth.contrastive_case(4242, B, N, h, h, K, H, H, list(range(16, 21)))
    # student features: a fixed direction per image half so early rows oppose late rows
    d = torch.nn.functional.normalize(synth.t_normal(1, (N,), stream=9), dim=0)
    f_n = 0.05 * f_n
    f_n[0] += d[None, :, None, None].expand(1, N, h, h)[0]
    f_n[1] -= d[None, :, None, None].expand(1, N, h, h)[0]
    ref_in = f_n.clone().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(ref_in, labels, l_po, f_o)
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], T)
    _, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], T)
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    pb = pixcon_prepare(fn_d.contiguous(memory_format=torch.channels_last), lab_d, lpo_d, fo_d, sort_by_label=False, fp16=True)
    loss_out, grad_a, stats = pixcon_loss_raw(pb, T, True, True, need_grad=True, row_stats=True, precision="f16")
    m = pb.meta_host()
    assert abs(loss_out[0].item() - ref.item()) / abs(ref.item()) < 1e-3
    torch.testing.assert_close(stats[0, :m.A].cpu().double(), neg, rtol=3e-3, atol=0)
    err = (grad_a[:m.A, :N].cpu().double() - da).abs().max().item() / da.abs().max().item()
    assert err < 3e-3, err


@pytest.mark.parametrize("dominant,B,h,H", [(9, 4, 33, 513), (3, 2, 21, 321), (None, 3, 33, 513)])
def test_fp16_planned_sweeps_vs_split_kernels_and_oracle(dominant, B, h, H):
    """The planned form of the fp16 path (pixcon_loss_f16p.hip: unit lists built on the device, sweep 1 skipping the tiles that
    lie wholly inside an anchor block's own label, the row maximum seeded with S_ii, padding rows corrected per unit, the self
    pair patched into the label word) against the fixed-split kernels it replaces (precision "f16_split", the A/B
    reference) and the fp32 oracle: one teacher class dominating (2/3 of all pairs are positives, whole anchor blocks carry
    one label - the skipping case) and the uniform case.  The plan itself is read back: with a dominant label sweep 1 must
    have dropped tiles."""
    from ucd_amd import hip
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare
    N, K = 256, 16
    f_n, f_o, l_po, labels = synth.contrastive_case(5150 + h, B, N, h, h, K, H, H, list(range(16, 21)))
    if dominant is not None:
        l_po[:, dominant] += 6.0
    prep = OC.pre_contrastive_pixel(f_n, labels, l_po, f_o)
    ref = OC.pixcon_loss(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    _, da, neg, G, num = OC.pixcon_loss_backward(prep["a"], prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last)
    out = {}
    for sort in (True, False):
        pb = pixcon_prepare(fn_d, lab_d, lpo_d, fo_d, sort_by_label=sort, fp16=True)
        m = pb.meta_host()
        for prec in ("f16", "f16_split"):
            loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True, precision=prec)
            if prec == "f16":
                ws = hip.workspace(hip.load().ucd_pixcon_loss_workspace_bytes(pb.BHW, pb.N, pb.K), fn_d.device, "pixloss")
                hdr = ws[:32].view(torch.int32).cpu().tolist()                      # ctr1 ctr2 U1 U2 CH1 CH2 nblk
                nblk, ntiles = (m.A + 127) // 128, m.Cpad // 32
                assert hdr[6] == nblk and hdr[0] >= hdr[2] > 0 and hdr[1] >= hdr[3] > 0          # every unit was drawn
                tiles1 = hdr[2] * hdr[4]                                            # upper bound of sweep-1 tile steps
                if sort and dominant is not None:
                    assert tiles1 < 0.8 * nblk * ntiles, (tiles1, nblk * ntiles)     # pure-positive tiles were skipped
                if not sort:
                    assert tiles1 >= nblk * ntiles                                  # nothing to skip without the grouping
            g_pix = torch.zeros(B * h * h, N, device=fn_d.device)
            g_pix[pb.anchor_pix[:m.A].long()] = grad_a[:m.A, :N].float()
            s_pix = torch.zeros(3, B * h * h, device=fn_d.device)
            s_pix[:, pb.anchor_pix[:m.A].long()] = stats[:, :m.A]
            out[(prec, sort)] = (loss_out[0].item(), g_pix, s_pix)
            assert abs(loss_out[0].item() - ref.item()) / abs(ref.item()) < 1e-3, (prec, sort)
            if not sort:                                                            # oracle rows are in pixel order
                torch.testing.assert_close(stats[0, :m.A].cpu().double(), neg, rtol=2e-3, atol=1e-6)
                torch.testing.assert_close(stats[1, :m.A].cpu().double(), num.double(), rtol=0, atol=0)
                err = (grad_a[:m.A, :N].cpu().double() - da).abs().max().item() / da.abs().max().item()
                assert err < 2e-3, (prec, err)
    for sort in (True, False):
        (l1, g1, s1), (l2, g2, s2) = out[("f16", sort)], out[("f16_split", sort)]
        assert abs(l1 - l2) / abs(l2) < 2e-4
        assert ((g1 - g2).norm() / g2.norm()).item() < 1e-3
        torch.testing.assert_close(s1[0], s2[0], rtol=1e-3, atol=1e-6)               # negative sums per anchor
        torch.testing.assert_close(s1[2], s2[2], rtol=1e-3, atol=1e-5)               # per-row losses
    assert ((out[("f16", True)][1] - out[("f16", False)][1]).norm() / out[("f16", False)][1].norm()).item() < 1e-3


def test_full_size_invariants_b24_513():
    """BASELINE.json's full per-GPU shape (B = 24, 513x513 -> 26136 pixels, up to 26136 x 52272 pairs), where the oracle's
    A x C matrices (5.5 GB each) do not fit a test: size-independent properties instead.  (1) The loss and the gradient do
    not depend on the order of the contrast rows (label-sorted vs pixel order).  (2) The exact-fp32 MFMA path and the fp16
    path agree to the performance mode's 1e-3.  (3) The row sums the kernels report reproduce the loss:
    loss = mean over valid anchors of the per-row losses."""
    from ucd_amd.contrastive import pixcon_loss_raw, pixcon_prepare
    B, N, h, K, H = 24, 256, 33, 16, 513
    f_n, f_o, l_po, labels = synth.contrastive_case(77, B, N, h, h, K, H, H, list(range(16, 21)))
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    fn_d = fn_d.contiguous(memory_format=torch.channels_last)
    res = {}
    for prec in ("f32", "f16"):
        for sort in (False, True):
            pb = pixcon_prepare(fn_d, lab_d, lpo_d, fo_d, sort_by_label=sort, fp16=(prec == "f16"))
            loss_out, grad_a, stats = pixcon_loss_raw(pb, 0.07, True, True, need_grad=True, row_stats=True, precision=prec)
            m = pb.meta_host()
            # scatter the anchor-row gradients back to pixels so that both orders are comparable
            g_pix = torch.zeros(B * h * h, N, device=fn_d.device)
            g_pix[pb.anchor_pix[:m.A].long()] = grad_a[:m.A, :N].float()
            res[(prec, sort)] = (loss_out[0].item(), g_pix, m.A, m.Co)
    A, Co = res[("f32", False)][2:]
    assert A > 5000 and Co > 5000, (A, Co)                       # a real full-size problem, not a degenerate one
    l32u, g32u = res[("f32", False)][:2]
    l32s, g32s = res[("f32", True)][:2]
    l16s, g16s = res[("f16", True)][:2]
    l16u, g16u = res[("f16", False)][:2]
    assert abs(l32u - l32s) / abs(l32u) < 1e-5
    assert ((g32u - g32s).norm() / g32u.norm()).item() < 1e-4
    assert abs(l16s - l32s) / abs(l32s) < 1e-3 and abs(l16u - l32u) / abs(l32u) < 1e-3
    assert ((g16s - g32s).norm() / g32s.norm()).item() < 2e-3
    assert ((g16u - g16s).norm() / g16s.norm()).item() < 2e-3


@pytest.mark.parametrize("name", PIXCON)
def test_pixelconlossv2_plain_tensor_call(name):
    """The reference's literal call shape (train.py:115-116): ``a, c, la, lc, P = pre_contractive_pixel(...)`` then
    ``PixelConLossV2()(a, c, la, lc, P)`` with five plain tensors - (i) the unpacked tensors of this package's own prep
    (found through the anchors tensor: fused kernel), (ii) foreign tensors holding the same values (``ucd_pixcon_loss_given_p``
    reads the materialised P), (iii) foreign tensors built by the ORACLE's prep, with and without P."""
    from ucd_amd.contrastive import PixelConLossV2, pre_contractive_pixel
    g = load_golden(f"pixcon_{name}.npz")
    f_n, f_o, l_po, labels = _case(g)
    fn_d, fo_d, lpo_d, lab_d = _to_dev(f_n, f_o, l_po, labels)
    crit = PixelConLossV2(temperature=0.07)
    # (i)
    x = fn_d.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a, c, la, lc, P = pre_contractive_pixel(x, lab_d, l_po=lpo_d, f_o=fo_d)
    loss = crit(a, c, la, lc, P)
    assert abs(loss.item() - float(g["loss"])) / abs(float(g["loss"])) < 1e-4
    loss.backward()
    assert_matches_compact(g, "grad_f_n", x.grad.cpu().numpy(), rtol=1e-3, atol=1e-7)
    a2, c2, la2, lc2, _ = pre_contractive_pixel(x.detach(), lab_d, l_po=lpo_d, f_o=fo_d)
    assert crit(a2, c2, la2, lc2).item() == pytest.approx(float(g["loss_noP"]), rel=1e-4)
    # (ii) same values, different tensor objects: nothing to look up
    af = a.detach().clone().requires_grad_(True)
    lf = crit(af, c.clone(), la.clone(), lc.clone(), P.clone())
    assert abs(lf.item() - float(g["loss"])) / abs(float(g["loss"])) < 1e-4
    lf.backward()
    # (iii) the oracle's tuple moved to the device
    ref_in = f_n.clone().requires_grad_(True)
    prep = OC.pre_contrastive_pixel(ref_in, labels, l_po, f_o)
    ao = prep["a"].detach().clone().requires_grad_(True)
    ref = OC.pixcon_loss(ao, prep["c"], prep["la"], prep["lc"], prep["P"], 0.07)
    ref.backward()
    ad = prep["a"].detach().to(fn_d.device).requires_grad_(True)
    args = [prep["c"].to(fn_d.device), prep["la"].to(fn_d.device), prep["lc"].to(fn_d.device)]
    lo = crit(ad, *args, prep["P"].to(fn_d.device))
    lo.backward()
    assert abs(lo.item() - ref.item()) / abs(ref.item()) < 1e-4
    err = (ad.grad.cpu() - ao.grad).abs().max().item() / ao.grad.abs().max().item()
    assert err < 1e-3, err
    err2 = (af.grad.cpu() - ao.grad).abs().max().item() / ao.grad.abs().max().item()
    assert err2 < 1e-3, err2
    ref_nop = OC.pixcon_loss(prep["a"].detach(), prep["c"], prep["la"], prep["lc"], None, 0.07)
    assert crit(ad.detach(), *args).item() == pytest.approx(ref_nop.item(), rel=1e-4)


def test_v1_pixelconloss_on_the_hip_path():
    """the dead-file PixelConLoss (utils/loss_new.py:359-400) is the V2 kernel's special case contrast = anchors,
    P = 1, no row-max shift - run that way through the C ABI and compared with the goldens captured from the reference."""
    import torch.nn.functional as F
    from ucd_amd.contrastive import PixelConLoss
    g = load_golden("v1_losses.npz")
    seed, n, d = [int(v) for v in g["cfg"]]
    f = F.normalize(synth.t_normal(seed, (n, d), stream=1), dim=1)
    lab = torch.from_numpy(synth.randint(seed, (n,), 0, 5, stream=2))
    dev = torch.device("cuda:0")
    fd, ld = f[:, None, :].to(dev), lab.to(dev)
    assert PixelConLoss(temperature=0.07)(fd, ld).item() == pytest.approx(float(g["pixcon_T007"]), rel=1e-4)
    assert PixelConLoss()(fd, ld).item() == pytest.approx(float(g["pixcon_T1"]), rel=1e-4)
    with pytest.raises(NotImplementedError):
        PixelConLoss()(fd.clone().requires_grad_(True), ld).backward()
