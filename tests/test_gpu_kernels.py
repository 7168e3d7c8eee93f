"""GPU parity tests of the individual HIP kernels (through the C-ABI) against the CPU oracle
/ stock torch-CPU fp32 ops on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import loss_ref, metrics_ref
from oracle.seeded import seeded_labels, seeded_tensor

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from kurosiwo_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2.5e-2


# ------------------------------------------------------------------ MFMA conventions
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_selftest_mma(dev, dtype):
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT, stream_ptr
    kc = 32 if dtype == torch.bfloat16 else 16
    a = seeded_tensor("mma.a", (16, kc)).to(dtype)
    b = seeded_tensor("mma.b", (16, kc)).to(dtype)       # asymmetric B (guide §3)
    c = torch.zeros((16, 16), dtype=torch.float32, device=dev)
    ad, bd = a.to(dev), b.to(dev)
    _lib.check(_lib.load().ksmi_selftest_mma(ad.data_ptr(), bd.data_ptr(), c.data_ptr(), DT[dtype], stream_ptr()))
    ref = a.float() @ b.float().t()
    assert (c.cpu() - ref).abs().max() < 1e-4 * ref.abs().max()


def test_selftest_tr16(dev):
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import stream_ptr
    inp = torch.arange(256, dtype=torch.int16, device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    _lib.check(_lib.load().ksmi_selftest_tr16(inp.data_ptr(), out.data_ptr(), stream_ptr()))
    got = out.cpu().numpy().reshape(64, 4)
    lane = np.arange(64)[:, None]
    j = np.arange(4)[None, :]
    exp = (lane & 15) + j * 16 + (lane >> 4) * 64
    assert (got == exp).all(), got[:20]


# ------------------------------------------------------------------ loss / metrics / optimiser
@pytest.mark.parametrize("shape,scale,pinv", [((1, 3, 2, 2), 1.0, 0.25), ((2, 3, 16, 16), 2.0, 0.05), ((3, 3, 224, 224), 4.0, 0.3)])
@pytest.mark.parametrize("with_dice", [True, False])
def test_loss_forward_backward(dev, shape, scale, pinv, with_dice):
    from kurosiwo_amd.loss import BCEandDiceLoss, CrossEntropyLoss
    x = seeded_tensor("kl.logits" + str(shape), shape) * scale
    t = seeded_labels("kl.labels" + str(shape), (shape[0],) + shape[2:], p_invalid=pinv)
    t[0, 0, 0] = 1
    crit = BCEandDiceLoss(CLASS_WEIGHTS, 3, True) if with_dice else CrossEntropyLoss(CLASS_WEIGHTS, 3)
    xd = x.to(dev).requires_grad_(True)
    loss = crit(xd, t.to(dev))
    (2.5 * loss).backward()
    r = loss_ref.ce_dice_forward(x.numpy(), t.numpy(), CLASS_WEIGHTS, with_dice=with_dice, with_grad=True)
    assert abs(float(loss) - r["total"]) < 2e-5 * max(1.0, abs(r["total"]))
    g = xd.grad.cpu().numpy() / 2.5
    assert np.abs(g - r["grad"]).max() < 2e-5 * np.abs(r["grad"]).max() + 1e-9


def test_loss_golden_kat(dev, golden_dir):
    import os
    from kurosiwo_amd.loss import BCEandDiceLoss
    gold = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    x = torch.tensor([[[[1, -.5], [.25, 2]], [[0, .5], [-1, .5]], [[-1, 1.5], [.75, -2]]]], dtype=torch.float32)
    t = torch.tensor([[[0, 2], [3, 1]]], dtype=torch.int64)
    xd = x.to(dev).requires_grad_(True)
    crit = BCEandDiceLoss([1.0, 1.0, 1.0], 3, True)
    loss = crit(xd, t.to(dev))
    loss.backward()
    assert abs(float(loss) - float(gold["kat.unit.total"])) < 2e-6
    assert np.abs(xd.grad.cpu().numpy() - gold["kat.unit.grad"]).max() < 2e-7


def test_argmax_confusion(dev):
    from kurosiwo_amd.metrics import ConfusionMetrics
    x = seeded_tensor("cm.logits", (3, 3, 64, 48))
    x[0, :, :4, :4] = 0.5                      # ties -> lowest index
    t = seeded_labels("cm.labels", (3, 64, 48), p_invalid=0.2)
    cmx = ConfusionMetrics(dev)
    pred = cmx.update(x.to(dev), t.to(dev), return_predictions=True)
    cmx.update(x.to(dev), t.to(dev))
    ref_pred = metrics_ref.argmax_lowest_index(x.numpy())
    assert (pred.cpu().numpy() == ref_pred).all()
    ref_cm = metrics_ref.confusion_matrix(ref_pred, t.numpy())
    assert (cmx.cm.cpu().numpy() == 2 * ref_cm).all()
    got, ref = cmx.compute(), metrics_ref.metrics_from_cm(2 * ref_cm)
    for k in ("accuracy", "precision", "recall", "f1", "iou"):
        assert np.abs(got[k].numpy() - ref[k]).max() < 1e-12
    assert abs(float(got["miou"]) - ref["miou"]) < 1e-12


# ------------------------------------------------------------------ convolution family
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    dict(B=2, H=32, W=32, cs=[32], N=32),
    dict(B=1, H=48, W=32, cs=[32, 32, 64], N=32),
    dict(B=2, H=16, W=16, cs=[64, 64, 128], N=64),
    dict(B=3, H=8, W=8, cs=[16, 8], N=24),
    dict(B=2, H=2, W=2, cs=[256], N=256),
    dict(B=1, H=14, W=14, cs=[128], N=128),
])
def test_conv3x3_forward_dgrad_wgrad(dev, dtype, cfg):
    from kurosiwo_amd import functional as Fk
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"conv{B}{H}{W}{cs}{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    K = sum(cs)
    w = seeded_tensor(tag + ".w", (N, K, 3, 3)) * (2.0 / (K * 9)) ** 0.5
    bias = seeded_tensor(tag + ".b", (N,)) * 0.1
    dy = seeded_tensor(tag + ".dy", (B, N, H, W))
    q = (lambda t: t.to(dtype).float())          # quantise the operands the way the kernel sees them
    xq = [q(x) for x in xs]
    wq, dyq = q(w), q(dy)
    xc = torch.cat(xq, 1).requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y_ref = F.conv2d(xc, wr, bias, padding=1)
    y_ref.backward(dyq)
    xd = [Fk.to_nhwc(x.to(dev), dtype) for x in xs]
    y, stats = Fk.conv3x3(xd, w.to(dev), bias.to(dev), want_stats=True)
    tol = _tol(dtype)
    yn = Fk.to_nchw(y).cpu()
    assert (yn - y_ref.detach()).abs().max() < tol * y_ref.abs().max()
    s = stats.sum(0).cpu()
    # the statistics rows describe the STORED output (round 5: what BatchNorm normalises is the tensor in memory, bf16-rounded in the
    # performance mode; sums of the fp32 accumulators fed a normalisation of values nobody reads and made bf16 training spike)
    yd = yn.double()
    assert (s[0, :N].double() - yd.sum((0, 2, 3))).abs().max() < 2e-5 * max(1.0, float(yd.abs().sum((0, 2, 3)).max()))
    assert (s[1, :N].double() - (yd ** 2).sum((0, 2, 3))).abs().max() < 2e-5 * float((yd ** 2).sum((0, 2, 3)).max())
    # ... and the stored output is the reference's to the rounding of the activation type
    assert (s[0, :N] - y_ref.detach().sum((0, 2, 3))).abs().max() < (5e-3 if dtype == torch.bfloat16 else 1e-3) * max(1.0, float(y_ref.abs().sum((0, 2, 3)).max()))
    dyd = Fk.to_nhwc(dy.to(dev), dtype)
    dxs = Fk.conv3x3_dgrad(dyd, w.to(dev), cs)
    dx = torch.cat([Fk.to_nchw(t).cpu() for t in dxs], 1)
    assert (dx - xc.grad).abs().max() < tol * xc.grad.abs().max()
    dw = Fk.conv3x3_wgrad(xd, dyd).cpu()
    assert (dw - wr.grad).abs().max() < tol * wr.grad.abs().max()


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=56, W=56, cs=[64], N=128),            # TW=8 patches, 64-column tiles, several splits
    dict(B=3, H=28, W=28, cs=[128, 32], N=64),        # TW=32 patches with masked columns, odd chunk count
    dict(B=2, H=14, W=14, cs=[256], N=512),           # 14x14 maps, 8 x 8 tiles
    dict(B=1, H=112, W=112, cs=[32], N=64),           # single chunk: 2 x 2 wave layout, 64 columns
    dict(B=1, H=224, W=224, cs=[32, 32, 64, 32], N=32),   # level-0 shape: virtual concat, 32 columns
    dict(B=2, H=33, W=19, cs=[64, 32], N=48),         # ragged map, N not a multiple of 32
    dict(B=1, H=40, W=72, cs=[32], N=32, aff=True),   # fused BN-apply + ReLU operand transformed in LDS
    dict(B=2, H=28, W=28, cs=[64], N=64, aff=True),
    dict(B=2, H=48, W=40, cs=[16], N=16),             # round 5: partial chunks (zero-page granules, the reducer skips the pad rows)
    dict(B=1, H=33, W=50, cs=[16], N=32, aff=True),
    dict(B=2, H=24, W=24, cs=[32, 16], N=16),
    dict(B=1, H=30, W=26, cs=[8], N=16),
    dict(B=1, H=20, W=44, cs=[16, 16, 16], N=16),     # FC-Siam conv12d: three 16-channel sources
    dict(B=2, H=40, W=36, cs=[64], N=2, dyC=8),       # 2- / 3-class heads: d out with a channel stride of 8
    dict(B=1, H=28, W=50, cs=[256], N=3, dyC=8),
])
def test_wgrad3_channel_owner_kernel(dev, cfg):
    """csrc/wgrad3.hip (bf16 3x3 s1 weight gradient, channel-owner tiling) vs F.conv2d's weight gradient on CPU."""
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"wg3{B}{H}{W}{cs}{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    dy = seeded_tensor(tag + ".dy", (B, N, H, W))
    q = (lambda t: t.to(dtype).float())
    K = sum(cs)
    aff = None
    xq = torch.cat([q(x) for x in xs], 1)
    if cfg.get("aff"):
        sc, sh = 1.0 + 0.3 * seeded_tensor(tag + ".sc", (K,)), 0.2 * seeded_tensor(tag + ".sh", (K,))
        xq = q(torch.relu(xq * sc[None, :, None, None] + sh[None, :, None, None]))
        aff = (sc.to(dev), sh.to(dev), 1)
    wr = torch.zeros((N, K, 3, 3)).requires_grad_(True)
    F.conv2d(xq, wr, None, padding=1).backward(q(dy))
    xd = [Fk.to_nhwc(x.to(dev), dtype) for x in xs]
    import ctypes as C
    from kurosiwo_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(4096)
    lib.ksmi_last_kernels(buf, 4096)
    if cfg.get("dyC"):
        dyp = torch.zeros((B, cfg["dyC"], H, W)); dyp[:, :N] = dy
        dw = Fk.conv3x3_wgrad(xd, Fk.to_nhwc(dyp.to(dev), dtype), affine=aff, n_real=N).cpu()
    else:
        dw = Fk.conv3x3_wgrad(xd, Fk.to_nhwc(dy.to(dev), dtype), affine=aff).cpu()
    assert lib.ksmi_last_kernels(buf, 4096) and "wgrad3_kernel" in buf.value.decode(), buf.value.decode()
    err = float((dw - wr.grad).abs().max() / wr.grad.abs().max())
    assert err < 2e-3, err            # operands are identical bf16 values; fp32 accumulation order differs


@pytest.mark.parametrize("nst", ["2", "3", "4"])
@pytest.mark.parametrize("cfg", [
    dict(B=4, H=56, W=56, cs=[64], N=64, aff=True, wgs="3"),       # 112 patches on 3 workgroups: the ring wraps ~12 times
    dict(B=2, H=33, W=19, cs=[64, 32], N=48, wgs="2"),             # ragged map (masked halo / columns read the zero page), 2 K tiles
    dict(B=3, H=40, W=72, cs=[32], N=32, aff=True, wgs="5"),       # single chunk (2 x 2 waves), uneven split tail
    dict(B=1, H=224, W=224, cs=[32, 32, 64, 32], N=32, wgs="16"),  # level-0 shape: virtual concat, 4 K tiles x 4 splits
])
def test_wgrad3_ring_depths(dev, cfg, nst):
    """The LDS ring of csrc/wgrad3.hip at every depth (KSMI_WGRAD3_NST) with the grid shrunk (KSMI_WGRAD3_WGS) so that every
    workgroup walks many patches: counted-vmcnt waits, stage reuse after the barrier, the drain at the end of a split.  The result
    is the same fixed-order sum for every depth, so the three depths must agree BITWISE with each other and match conv2d."""
    import os
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"wg3r{B}{H}{W}{cs}{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    dy = seeded_tensor(tag + ".dy", (B, N, H, W))
    q = (lambda t: t.to(dtype).float())
    K = sum(cs)
    aff = None
    xq = torch.cat([q(x) for x in xs], 1)
    if cfg.get("aff"):
        sc, sh = 1.0 + 0.3 * seeded_tensor(tag + ".sc", (K,)), 0.2 * seeded_tensor(tag + ".sh", (K,))
        xq = q(torch.relu(xq * sc[None, :, None, None] + sh[None, :, None, None]))
        aff = (sc.to(dev), sh.to(dev), 1)
    wr = torch.zeros((N, K, 3, 3)).requires_grad_(True)
    F.conv2d(xq, wr, None, padding=1).backward(q(dy))
    xd = [Fk.to_nhwc(x.to(dev), dtype) for x in xs]
    dyd = Fk.to_nhwc(dy.to(dev), dtype)

    def run(depth):
        from kurosiwo_amd import _lib
        with _lib.knobs(KSMI_WGRAD3_NST=depth, KSMI_WGRAD3_WGS=cfg["wgs"]):      # (run-time knobs: include/ksmi.h ksmi_set_knob)
            return Fk.conv3x3_wgrad(xd, dyd, affine=aff).cpu()

    dw = run(nst)
    err = float((dw - wr.grad).abs().max() / wr.grad.abs().max())
    assert err < 2e-3, err
    assert torch.equal(dw, run(nst)), "run-to-run difference: a stage was read before its DMA landed"
    assert torch.equal(dw, run("2")), "ring depth changed the sum"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_fused_affine_relu_operand(dev, dtype):
    from kurosiwo_amd import functional as Fk
    B, H, W, Cc, N = 2, 24, 40, 32, 32
    x = seeded_tensor("aff.x", (B, Cc, H, W))
    w = seeded_tensor("aff.w", (N, Cc, 3, 3)) * 0.1
    sc, sh = 1.0 + 0.3 * seeded_tensor("aff.sc", (Cc,)), 0.2 * seeded_tensor("aff.sh", (Cc,))
    dy = seeded_tensor("aff.dy", (B, N, H, W))
    q = (lambda t: t.to(dtype).float())
    xa = q(torch.relu(q(x) * sc[None, :, None, None] + sh[None, :, None, None]))
    wr = q(w).requires_grad_(True)
    y_ref = F.conv2d(xa, wr, None, padding=1)       # zero padding applies AFTER the affine+relu
    y_ref.backward(q(dy))
    xd = Fk.to_nhwc(x.to(dev), dtype)
    aff = (sc.to(dev), sh.to(dev), 1)
    y, _ = Fk.conv3x3([xd], w.to(dev), None, affine=aff)
    assert (Fk.to_nchw(y).cpu() - y_ref.detach()).abs().max() < _tol(dtype) * y_ref.abs().max()
    dw = Fk.conv3x3_wgrad([xd], Fk.to_nhwc(dy.to(dev), dtype), affine=aff).cpu()
    assert (dw - wr.grad).abs().max() < _tol(dtype) * wr.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 7, 5, 32), (2, 1, 1, 256)])
def test_deconv2x2(dev, dtype, shape):
    from kurosiwo_amd import functional as Fk
    B, H, W, Cc = shape
    x = seeded_tensor("dc.x" + str(shape), (B, Cc, H, W))
    w = seeded_tensor("dc.w" + str(shape), (Cc, Cc, 2, 2)) * (1.0 / Cc) ** 0.5
    b = seeded_tensor("dc.b" + str(shape), (Cc,)) * 0.1
    dy = seeded_tensor("dc.dy" + str(shape), (B, Cc, 2 * H, 2 * W))
    q = (lambda t: t.to(dtype).float())
    xr, wr = q(x).requires_grad_(True), q(w).requires_grad_(True)
    y_ref = F.conv_transpose2d(xr, wr, b, stride=2)
    y_ref.backward(q(dy))
    xd = Fk.to_nhwc(x.to(dev), dtype)
    y = Fk.deconv2x2(xd, w.to(dev), b.to(dev))
    tol = _tol(dtype)
    assert (Fk.to_nchw(y).cpu() - y_ref.detach()).abs().max() < tol * y_ref.abs().max()
    dx, dw = Fk.deconv2x2_backward(xd, Fk.to_nhwc(dy.to(dev), dtype), w.to(dev))
    assert (Fk.to_nchw(dx).cpu() - xr.grad).abs().max() < tol * xr.grad.abs().max()
    assert (dw.cpu() - wr.grad).abs().max() < tol * wr.grad.abs().max()


def test_adam_and_sgd_match_reference_formulas(dev):
    """ksmi_adam_step / ksmi_adamw_step / ksmi_sgd_step on a flat arena vs torch.optim.Adam / AdamW / SGD (CPU) over 4 steps
    (the three optimiser branches of training/change_detection_trainer.py:45-66)."""
    from kurosiwo_amd.optim import FusedAdam, FusedAdamW, FusedSGD
    n = 10007
    p0 = seeded_tensor("opt.p", (n,))
    grads = [seeded_tensor(f"opt.g{i}", (n,)) * (10.0 ** (-i)) for i in range(4)]
    for kind in ("adam", "adamw", "sgd"):
        pr = p0.clone().requires_grad_(True)
        ref = (torch.optim.Adam([pr], lr=1e-3) if kind == "adam" else
               torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.05) if kind == "adamw" else
               torch.optim.SGD([pr], lr=6e-4, momentum=0.99, weight_decay=1e-5))
        pd = torch.nn.Parameter(p0.clone().to(dev))
        opt = (FusedAdam([pd], lr=1e-3) if kind == "adam" else FusedAdamW([pd], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.05) if kind == "adamw" else
               FusedSGD([pd], lr=6e-4, momentum=0.99, weight_decay=1e-5))
        for g in grads:
            pr.grad = g.clone()
            ref.step()
            pd.grad = g.clone().to(dev)
            opt.step()
        assert (pd.detach().cpu() - pr.detach()).abs().max() < 2e-6, kind


def test_sar_preprocess_matches_dataset_pipeline():
    """clamp -> nan_to_num -> Normalize of dataset/Dataset.py:164-168,193-198."""
    from kurosiwo_amd.data import preprocess_gpu
    g = torch.Generator().manual_seed(5)
    x = torch.rand((3, 2, 64, 48), generator=g) * 0.4 - 0.05
    x[0, 0, 3, 4] = float("nan")
    x[1, 1, 0, 0] = float("inf")
    x[2, 0, 5, 5] = -float("inf")
    mean, std, cl = [0.0953, 0.0264], [0.0427, 0.0215], 0.15
    ref = torch.nan_to_num(torch.clamp(x, min=0.0, max=cl), cl)
    ref = (ref - torch.tensor(mean).view(1, 2, 1, 1)) / torch.tensor(std).view(1, 2, 1, 1)
    out = preprocess_gpu(x.cuda(), mean, std, cl).cpu()
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,W", [(4, 24, 20), (8, 112, 112)])      # 3 splits (4 slab lanes) / 196 splits (16 slab lanes, two accumulators)
def test_up_weight_gradient_c64(dev, B, H, W):
    """the weight gradient of the level-0 Ups (C = 64: a 128-column tile is exactly one 2C run of the depth row) on ksmi_up_wgrad"""
    import torch.nn.functional as F
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import stream_ptr
    lib = _lib.load()
    Cc = 64
    assert lib.ksmi_up_wgrad_supported(B, H, W, Cc, 1) == 1 and lib.ksmi_up_gemm_supported(B, H, W, Cc, 1) == 0
    torch.manual_seed(64)
    x = (torch.randn(B, Cc, H, W, device=dev) * 0.5).bfloat16()
    dy = (torch.randn(B, Cc, 2 * H, 2 * W, device=dev) * 0.5).bfloat16()
    wr = torch.zeros(Cc, Cc, 2, 2, device=dev, requires_grad=True)
    F.conv_transpose2d(x.float(), wr, None, stride=2).backward(dy.float())
    ws = torch.empty(lib.ksmi_up_wgrad_workspace(B, H, W, Cc), dtype=torch.uint8, device=dev)
    gw = torch.full((Cc, Cc, 2, 2), 3.0, device=dev)
    # (the NHWC copies are held in variables: a temporary's block goes back to the caching allocator the moment data_ptr() returns and
    # the next temporary may be carved from it -- the launch then reads the wrong tensor, depending on what earlier tests left cached)
    x_nhwc, dy_nhwc = x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    _lib.check(lib.ksmi_up_wgrad(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), ws.data_ptr(), gw.data_ptr(), 0, B, H, W, Cc, stream_ptr()), "wgrad")
    assert float((gw - wr.grad).abs().max() / wr.grad.abs().max()) < 2e-3
    first = gw.clone()
    _lib.check(lib.ksmi_up_wgrad(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), ws.data_ptr(), gw.data_ptr(), 0, B, H, W, Cc, stream_ptr()), "wgrad")
    assert torch.equal(gw, first)                 # fixed summation order


@pytest.mark.parametrize("B,H,W,Cc", [(2, 28, 28, 128), (3, 14, 14, 256), (2, 7, 9, 512), (32, 56, 56, 128)])
def test_up_convtranspose_as_token_gemms(dev, B, H, W, Cc):
    """ksmi_up_forward / ksmi_up_dgrad / ksmi_up_wgrad (ConvTranspose2d(k2, s2) of `up`, models/snunet.py:32-46, as token GEMMs over
    "depth rows" of the NHWC output; csrc/gemm2.hip) against torch.nn.functional.conv_transpose2d and its autograd on the same
    bf16-rounded operands."""
    import ctypes as C
    import torch.nn.functional as F
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import stream_ptr
    lib = _lib.load()
    assert lib.ksmi_up_gemm_supported(B, H, W, Cc, 1) == 1
    torch.manual_seed(B * 1000 + Cc)
    x = (torch.randn(B, Cc, H, W, device=dev) * 0.5).bfloat16()
    wt = torch.randn(Cc, Cc, 2, 2, device=dev) / (Cc ** 0.5)
    bias = torch.randn(Cc, device=dev)
    dy = (torch.randn(B, Cc, 2 * H, 2 * W, device=dev) * 0.5).bfloat16()
    # reference on the bf16-rounded weights (the kernel's operand), fp32 math
    wr = wt.bfloat16().float().requires_grad_(True)
    xr = x.float().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, bias, stride=2)
    yr.backward(dy.float())
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous()
    wb = torch.empty(4 * Cc * Cc, dtype=torch.bfloat16, device=dev)
    st = stream_ptr()
    _lib.check(lib.ksmi_up_pack_weight(wt.data_ptr(), wb.data_ptr(), Cc, st), "pack")
    # the batched form (one launch for all `up` weights of a plan) writes the same images
    import ctypes
    w2 = torch.randn(64, 64, 2, 2, device=dev)
    wb_b, wb2_b = torch.zeros_like(wb), torch.zeros(4 * 64 * 64, dtype=torch.bfloat16, device=dev)
    wb2 = torch.empty_like(wb2_b)
    _lib.check(lib.ksmi_up_pack_weight(w2.data_ptr(), wb2.data_ptr(), 64, st), "pack")
    _lib.check(lib.ksmi_up_pack_weights_batched((ctypes.c_void_p * 2)(wt.data_ptr(), w2.data_ptr()), (ctypes.c_void_p * 2)(wb_b.data_ptr(), wb2_b.data_ptr()),
                                                (ctypes.c_int * 2)(Cc, 64), 2, st), "pack_batched")
    assert torch.equal(wb_b, wb) and torch.equal(wb2_b, wb2)
    y = torch.empty(B, 2 * H, 2 * W, Cc, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.ksmi_up_forward(x_nhwc.data_ptr(), wb.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, Cc, st), "fwd")
    ref = yr.detach().permute(0, 2, 3, 1)
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 1e-2
    dx = torch.full((B, H, W, Cc), 7.0, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.ksmi_up_dgrad(dy_nhwc.data_ptr(), wb.data_ptr(), dx.data_ptr(), 0, B, H, W, Cc, st), "dgrad")
    refd = xr.grad.permute(0, 2, 3, 1)
    assert float((dx.float() - refd).abs().max() / refd.abs().max()) < 1e-2
    base = dx.clone()
    _lib.check(lib.ksmi_up_dgrad(dy_nhwc.data_ptr(), wb.data_ptr(), dx.data_ptr(), 1, B, H, W, Cc, st), "dgrad+=")
    assert float((dx.float() - (base.float() + refd)).abs().max() / refd.abs().max()) < 2e-2
    ws = torch.empty(lib.ksmi_up_wgrad_workspace(B, H, W, Cc), dtype=torch.uint8, device=dev)
    gw = torch.full((Cc, Cc, 2, 2), 3.0, device=dev)
    _lib.check(lib.ksmi_up_wgrad(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), ws.data_ptr(), gw.data_ptr(), 0, B, H, W, Cc, st), "wgrad")
    assert float((gw - wr.grad).abs().max() / wr.grad.abs().max()) < 2e-3
    _lib.check(lib.ksmi_up_wgrad(x_nhwc.data_ptr(), dy_nhwc.data_ptr(), ws.data_ptr(), gw.data_ptr(), 1, B, H, W, Cc, st), "wgrad+=")
    assert float((gw - 2 * wr.grad).abs().max() / wr.grad.abs().max()) < 4e-3


@pytest.mark.parametrize("shape", [(2, 64, 56, 56), (1, 16, 31, 45), (3, 8, 7, 9)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_maxpool3x3s2_recorded_first_maximum(dev, dtype, shape):
    """ksmi_maxpool3x3s2_forward_idx / _backward_idx against torch.nn.functional.max_pool2d(3, 2, 1) and against the gather pair:
    the input is a ReLU output quantised to bf16 (zeros and equal positive values: ties everywhere), the gradient of a window has to go
    to the FIRST maximum in scan order."""
    from kurosiwo_amd import _lib, functional as Fk
    from kurosiwo_amd.functional import DT, stream_ptr
    lib = _lib.load()
    B, Cc, H, W = shape
    x = torch.relu(seeded_tensor(f"mp3.x{shape}", (B, Cc, H, W))).to(torch.bfloat16).float()
    x = (x * 4).round() / 4                                   # few distinct values: many equal maxima inside a window
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = seeded_tensor(f"mp3.dy{shape}", (B, Cc, Ho, Wo)).to(torch.bfloat16).float()
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    y_ref.backward(dy)
    xd = Fk.to_nhwc(x.to(dev), dtype)
    dyd = Fk.to_nhwc(dy.to(dev), dtype)
    y = torch.empty((B, Ho, Wo, Cc), dtype=dtype, device=dev)
    idx = torch.empty(y.numel(), dtype=torch.uint8, device=dev)
    _lib.check(lib.ksmi_maxpool3x3s2_forward_idx(xd.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, Cc, DT[dtype], stream_ptr()), "fwd_idx")
    assert torch.equal(Fk.to_nchw(y).cpu(), y_ref.detach())
    for acc in (0, 1):
        dx = torch.full((B, H, W, Cc), 0.5, dtype=dtype, device=dev)
        _lib.check(lib.ksmi_maxpool3x3s2_backward_idx(idx.data_ptr(), dyd.data_ptr(), dx.data_ptr(), acc, B, H, W, Cc, DT[dtype], stream_ptr()), "bwd_idx")
        dx2 = torch.full((B, H, W, Cc), 0.5, dtype=dtype, device=dev)
        _lib.check(lib.ksmi_maxpool3x3s2_backward(xd.data_ptr(), dyd.data_ptr(), dx2.data_ptr(), acc, B, H, W, Cc, DT[dtype], stream_ptr()), "bwd")
        assert torch.equal(dx, dx2)                           # the recorded route == the gather route, bit for bit
        ref = xr.grad + (0.5 if acc else 0.0)
        tol = 0.0 if dtype == torch.float32 else 2e-2 * float(ref.abs().max())
        assert (Fk.to_nchw(dx).cpu() - ref).abs().max() <= tol


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=48, W=40, cs=[32], N=32),                       # igemm3 (short K, weights in registers)
    dict(B=4, H=96, W=96, cs=[32, 32, 64], N=32),               # igemm4 <8, 2> (long K, 32 columns; >= 64 tiles so that it is chosen)
    dict(B=4, H=96, W=96, cs=[64, 64], N=128),                  # igemm4 <4, 4>
    dict(B=1, H=14, W=14, cs=[256], N=256),                     # igemm2 (tiny map: the tile kernel)
])
def test_conv_statistics_are_sums_over_the_stored_bf16_values(dev, cfg):
    """the BatchNorm statistics rows of every convolution epilogue are the sum and the sum of squares of the values AS
    STORED (bf16), not of the fp32 accumulators -- BatchNorm normalises the stored tensor (round 5 found loss excursions of 2.6 x ... 16 x
    in bf16 K-step runs when the two differed).  Direct check per kernel generation: reduce the stats rows and compare with fp64 sums over
    the output tensor the launch stored; the fp32-accumulator sums differ from those by the rounding noise this test would catch."""
    from kurosiwo_amd import functional as Fk
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"stats.{H}.{sum(cs)}.{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    w = seeded_tensor(tag + ".w", (N, sum(cs), 3, 3)) * (2.0 / (9 * sum(cs))) ** 0.5 * 3.0
    xd = [Fk.to_nhwc(x.to(dev), torch.bfloat16) for x in xs]
    out, stats = Fk.conv3x3(xd, w.to(dev), want_stats=True)
    torch.cuda.synchronize()
    y = out.float().double().reshape(-1, out.shape[-1])[:, :N]          # the stored values
    s1, s2 = y.sum(0), (y * y).sum(0)
    g1, g2 = stats[:, 0, :N].double().sum(0), stats[:, 1, :N].double().sum(0)
    # fp32 partial sums per workgroup row: relative 1e-5 of the absolute mass
    tol1 = 2e-5 * y.abs().sum(0) + 1e-6
    assert ((g1 - s1).abs() <= tol1).all(), float(((g1 - s1).abs() / tol1).max())
    assert ((g2 - s2).abs() <= 2e-5 * s2 + 1e-6).all(), float(((g2 - s2).abs() / (2e-5 * s2 + 1e-6)).max())
    # ... and the test has teeth: sums over the UNROUNDED convolution differ by more than that tolerance in a large share of the channels
    ref = torch.nn.functional.conv2d(torch.cat([x.to(torch.bfloat16).float() for x in xs], 1).double(), w.to(torch.bfloat16).double(), padding=1)
    r2 = (ref * ref).sum((0, 2, 3))
    assert ((r2.to(dev) - s2).abs() > 2e-5 * s2 + 1e-6).float().mean() >= 0.3
