"""GPU parity of the persistent long-K convolution kernel (csrc/igemm4.hip: halo ring + weight ring by counted LDS-DMA, one
statistics row per workgroup) against stock torch-CPU fp32 convolutions on the same bf16-quantised operands.
KSMI_IGEMM4_CUS shrinks the persistent grid so that every workgroup walks several tiles (ring wrap across tile boundaries, the
tail of the ring); KSMI_IGEMM4_NF forces the 64- / 128-column workgroup tile."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.seeded import seeded_tensor


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from kurosiwo_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def q(t):
    return t.to(torch.bfloat16).float()


def _variants(N):
    """(KSMI_IGEMM4_CUS, KSMI_IGEMM4_VAR) settings that apply to a layer with N output channels: the default choice on the full
    machine, then every workgroup shape (pixel groups, column fragments per wave) on a tiny grid (many tiles per workgroup)"""
    v = [("256", None)]
    if N % 128 == 0:
        v.append(("5", "4,4"))
    if N % 64 == 0:
        v += [("3", "8,4"), ("4", "4,2")]
    if N == 32:
        v.append(("3", "8,2"))
    return v


class _Env:
    """run-time knobs of the launchers (include/ksmi.h ksmi_set_knob): set for the block, back to the built-in defaults after it"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        from kurosiwo_amd import _lib
        for k, v in self.kv.items():
            _lib.set_knob(k, v)

    def __exit__(self, *a):
        from kurosiwo_amd import _lib
        for k in self.kv:
            _lib.set_knob(k, None)


CFGS = [
    dict(B=2, H=40, W=40, cs=[96], N=64),                     # three chunks, 64-column tile
    dict(B=1, H=33, W=50, cs=[32, 32, 64], N=128),            # virtual concat, ragged map, 128 columns
    dict(B=3, H=28, W=28, cs=[128], N=128, aff=True),         # fused BN-apply + ReLU operand, transformed in LDS
    dict(B=2, H=30, W=22, cs=[64], N=64, mask=True),          # ReLU-mask + BN-backward sums epilogue
    dict(B=2, H=14, W=14, cs=[256], N=256, mask=True),
    dict(B=2, H=24, W=24, cs=[64, 64], N=64, acc=True),       # dst += result
    dict(B=1, H=56, W=56, cs=[96], N=192),                    # three chunks; three column tiles of 64
    dict(B=4, H=16, W=16, cs=[160], N=64, aff=True),          # odd chunk count through the 3-slot rings
    dict(B=2, H=48, W=40, cs=[32, 64], N=32),                 # 32 output channels: 512-pixel patches, 64 x 32 wave tiles
    dict(B=1, H=64, W=64, cs=[64], N=32, mask=True),
    dict(B=2, H=48, W=40, cs=[256], N=2, ostride=8),          # 2-class head into a destination with channel stride 8 (ChangeFormer)
    dict(B=1, H=33, W=50, cs=[64, 32], N=3, ostride=8),
]


@pytest.mark.parametrize("cfg", CFGS)
def test_igemm4_conv3x3(dev, cfg):
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"ig4.{B}{H}{W}{cs}{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    K = sum(cs)
    w = seeded_tensor(tag + ".w", (N, K, 3, 3)) * (2.0 / (K * 9)) ** 0.5
    bias = seeded_tensor(tag + ".b", (N,)) * 0.1
    xq = torch.cat([q(x) for x in xs], 1)
    aff = None
    if cfg.get("aff"):
        sc, sh = 1.0 + 0.3 * seeded_tensor(tag + ".sc", (K,)), 0.2 * seeded_tensor(tag + ".sh", (K,))
        xq = q(torch.relu(xq * sc[None, :, None, None] + sh[None, :, None, None]))
        aff = (sc.to(dev), sh.to(dev), 1)
    y_ref = F.conv2d(xq, q(w), bias, padding=1)
    s0_ref, s1_ref = y_ref.sum((0, 2, 3)), (y_ref ** 2).sum((0, 2, 3))
    mask = None
    if cfg.get("mask"):
        m = q(seeded_tensor(tag + ".m", (B, N, H, W)))
        mean, rstd = 0.1 * seeded_tensor(tag + ".mm", (N,)), 1.0 + 0.2 * seeded_tensor(tag + ".mr", (N,)).abs()
        msc, msh = 1.0 + 0.3 * seeded_tensor(tag + ".ms", (N,)), 0.2 * seeded_tensor(tag + ".mh", (N,))
        keep = (m * msc[None, :, None, None] + msh[None, :, None, None]) > 0
        y_ref = torch.where(keep, y_ref, torch.zeros_like(y_ref))
        xh = (m - mean[None, :, None, None]) * rstd[None, :, None, None]
        s0_ref, s1_ref = y_ref.sum((0, 2, 3)), (y_ref * xh).sum((0, 2, 3))
        mask = (Fk.to_nhwc(m.to(dev), dtype), mean.to(dev), rstd.to(dev), msc.to(dev), msh.to(dev))
    old = q(seeded_tensor(tag + ".old", (B, N, H, W))) if cfg.get("acc") else None
    xd = [Fk.to_nhwc(x.to(dev), dtype) for x in xs]
    for cus, var in _variants(N):
        if cfg.get("ostride"):
            import ctypes as C
            from kurosiwo_amd import _lib
            lib = _lib.load()
            buf = C.create_string_buffer(4096)
            lib.ksmi_last_kernels(buf, 4096)
            outp = torch.full((B, H, W, cfg["ostride"]), 7.0, dtype=dtype, device=dev)
            with _Env(KSMI_IGEMM4_CUS="3", KSMI_IGEMM4_VAR="8,2"):      # (small maps: force the 512-pixel x 32-column variant, many tiles per workgroup)
                y, stats = Fk.conv3x3(xd, w.to(dev), bias.to(dev), want_stats=True, out=outp)
            torch.cuda.synchronize()
            assert lib.ksmi_last_kernels(buf, 4096) and "igemm4_kernel" in buf.value.decode(), buf.value.decode()
            yn = Fk.to_nchw(y).cpu()
            assert (yn[:, :N] - y_ref).abs().max() < 2.5e-2 * y_ref.abs().max()
            assert float(yn[:, N:].abs().max()) == 0.0
            s = stats.sum(0).cpu()
            assert (s[0, :N] - s0_ref).abs().max() < 1e-3 * max(1.0, float(y_ref.abs().sum((0, 2, 3)).max()))
            continue
        out = Fk.to_nhwc(old.to(dev), dtype) if old is not None else None
        with _Env(KSMI_IGEMM4_CUS=cus, KSMI_IGEMM4_VAR=var):
            y, stats = Fk.conv3x3(xd, w.to(dev), bias.to(dev), affine=aff, want_stats=True, mask=mask, out=out,
                                  accumulate=1 if old is not None else 0)
        yn = Fk.to_nchw(y).cpu()
        y_chk = y_ref + old if old is not None else y_ref
        assert (yn - y_chk).abs().max() < 2.5e-2 * y_chk.abs().max(), (cus, var)
        s = stats.sum(0).cpu()
        assert (s[0, :N] - s0_ref).abs().max() < 1e-3 * max(1.0, float(y_ref.abs().sum((0, 2, 3)).max())), (cus, var)
        assert (s[1, :N] - s1_ref).abs().max() < 2e-3 * max(1.0, float((y_ref ** 2).sum((0, 2, 3)).max())), (cus, var)
        if var is not None:     # the persistent kernel ran: one statistics row per pixel-axis workgroup (a tile kernel writes B * tiles rows)
            assert stats.shape[0] <= 8, (cus, var, stats.shape)


@pytest.mark.parametrize("cfg", [dict(B=2, H=40, W=36, cs=[32, 32, 32], N=32), dict(B=1, H=28, W=28, cs=[128, 128], N=128),
                                 dict(B=2, H=24, W=24, cs=[64], N=64, first=True)])
def test_igemm4_gate_epilogue(dev, cfg):
    """total gradient (+ old destination), ReLU gate of the block output, BatchNorm2-backward sums (ksmi_conv_desc.gate_src)"""
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    B, H, W, cs, N = cfg["B"], cfg["H"], cfg["W"], cfg["cs"], cfg["N"]
    tag = f"ig4g.{B}{H}{W}{cs}{N}"
    xs = [seeded_tensor(f"{tag}.x{i}", (B, c, H, W)) for i, c in enumerate(cs)]
    K = sum(cs)
    w = seeded_tensor(tag + ".w", (N, K, 3, 3)) * (2.0 / (K * 9)) ** 0.5
    old = q(seeded_tensor(tag + ".old", (B, N, H, W))) * 0.5
    outb = q(seeded_tensor(tag + ".out", (B, N, H, W)))
    z = q(seeded_tensor(tag + ".z", (B, N, H, W)))
    mean, rstd = 0.1 * seeded_tensor(tag + ".mm", (N,)), 1.0 + 0.2 * seeded_tensor(tag + ".mr", (N,)).abs()
    y = F.conv2d(torch.cat([q(x) for x in xs], 1), q(w), None, padding=1)
    first = cfg.get("first", False)
    tot = y if first else y + old
    g_ref = q(torch.where(outb > 0, tot, torch.zeros_like(tot)))
    zh = (z - mean[None, :, None, None]) * rstd[None, :, None, None]
    s0_ref, s1_ref = g_ref.sum((0, 2, 3)), (g_ref * zh).sum((0, 2, 3))
    xd = [Fk.to_nhwc(x.to(dev), dtype) for x in xs]
    gate = (Fk.to_nhwc(outb.to(dev), dtype), Fk.to_nhwc(z.to(dev), dtype), mean.to(dev), rstd.to(dev))
    for cus, var in _variants(N):
        dst = Fk.to_nhwc(old.to(dev), dtype)
        with _Env(KSMI_IGEMM4_CUS=cus, KSMI_IGEMM4_VAR=var):
            try:
                g, stats = Fk.conv3x3(xd, w.to(dev), None, want_stats=True, out=dst, accumulate=0 if first else 1, gate=gate)
            except Exception as e:      # the default choice leaves maps with < 64 patches to the tile kernel, which has no gate epilogue
                assert var is None and "gate epilogue is not available" in str(e)
                continue
        gn = Fk.to_nchw(g).cpu()
        assert (gn - g_ref).abs().max() < 2.5e-2 * g_ref.abs().max(), (cus, var)
        assert ((gn != 0) & (outb <= 0)).sum() == 0                      # the gate is exact
        s = stats.sum(0).cpu()
        assert (s[0, :N] - s0_ref).abs().max() < 2e-3 * max(1.0, float(g_ref.abs().sum((0, 2, 3)).max())), (cus, var)
        assert (s[1, :N] - s1_ref).abs().max() < 2e-3 * max(1.0, float((g_ref * zh).abs().sum((0, 2, 3)).max())), (cus, var)


def test_igemm4_residual_block_epilogues(dev):
    """ResidualBlock of models/changeformer.py:471-483 on the persistent kernel: relu(conv1(x) + b), then 0.1 * (conv2(r) + b) + x"""
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    B, H, W, E = 2, 32, 32, 128
    x = q(seeded_tensor("ig4r.x", (B, E, H, W)))
    w1 = seeded_tensor("ig4r.w1", (E, E, 3, 3)) * (2.0 / (E * 9)) ** 0.5
    w2 = seeded_tensor("ig4r.w2", (E, E, 3, 3)) * (2.0 / (E * 9)) ** 0.5
    b1, b2 = 0.1 * seeded_tensor("ig4r.b1", (E,)), 0.1 * seeded_tensor("ig4r.b2", (E,))
    r_ref = q(torch.relu(F.conv2d(x, q(w1), b1, padding=1)))
    y_ref = 0.1 * F.conv2d(r_ref, q(w2), b2, padding=1) + x
    xd = Fk.to_nhwc(x.to(dev), dtype)
    for cus, var in _variants(E):
        with _Env(KSMI_IGEMM4_CUS=cus, KSMI_IGEMM4_VAR=var):
            r, _ = Fk.conv3x3([xd], w1.to(dev), b1.to(dev), relu_out=1)
            y, _ = Fk.conv3x3([r], w2.to(dev), b2.to(dev), alpha=0.1, resid=xd)
        assert (Fk.to_nchw(r).cpu() - r_ref).abs().max() < 2.5e-2 * r_ref.abs().max(), (cus, var)
        assert (Fk.to_nchw(y).cpu() - y_ref).abs().max() < 2.5e-2 * y_ref.abs().max(), (cus, var)


def test_igemm4_matches_igemm2_bitwise_inputs(dev):
    """same descriptor with and without statistics: the output tensor must not depend on the statistics epilogue"""
    from kurosiwo_amd import functional as Fk
    dtype = torch.bfloat16
    x = Fk.to_nhwc(seeded_tensor("ig4.same.x", (2, 96, 32, 32)).to(dev), dtype)
    w = seeded_tensor("ig4.same.w", (64, 96, 3, 3)).to(dev) * 0.05
    y0, _ = Fk.conv3x3([x], w, None, want_stats=False)
    y1, st = Fk.conv3x3([x], w, None, want_stats=True)
    assert torch.equal(y0, y1)
    assert st is not None and torch.isfinite(st).all()


@pytest.mark.parametrize("switch", ["KSMI_IG4_NW4", "KSMI_IG4_DEEP", "KSMI_IG4_CHUNK"])
def test_opt_in_schedules_in_their_own_process(switch):
    """The round-5 variants (4-wave workgroups two per CU; 5-slot weight ring with four steps of lead; one barrier per chunk) are compiled into the library and
    selected by switches the launcher reads once per process: this file's cases run again in a child process with the switch on --
    the 32-column cases (plain, mask, gate epilogues; virtual concat; full machine and a 3-workgroup grid) are the ones they serve, the
    others must be unaffected -- plus a check that the switch really changed the geometry (statistics rows of a level-0 shape)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **{switch: "1"})
    probe = ("import sys, torch; sys.path.insert(0, %r); from kurosiwo_amd.runtime import make_conv, SrcSpec, conv_stats_rows; "
             "x = torch.empty(1, dtype=torch.bfloat16); "
             "d, t = make_conv([SrcSpec(x, 32), SrcSpec(x, 32), SrcSpec(x, 64)], [(x, 32, 0, 0, 32, 0)], x, None, None, 32, 224, 224, 224, 224, 3, 3, 1, 1, 32, torch.bfloat16); "
             "print('ROWS', conv_stats_rows(d, torch.bfloat16))" % root)
    rows = {}
    for tag, e in (("off", dict(os.environ)), ("on", env)):
        e.pop(switch, None) if tag == "off" else None
        out = subprocess.run([sys.executable, "-c", probe], env=e, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        rows[tag] = int(out.stdout.split("ROWS")[1].split()[0])
    if switch == "KSMI_IG4_NW4":
        assert rows["on"] > 256 >= rows["off"], rows            # two workgroups per CU: up to 512 persistent workgroups
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k", "not opt_in"],
                         env=env, capture_output=True, text=True, timeout=1800, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def _deconv_k4s2_via_phases(x, w, bias, dy, mask_t, dev):
    """ConvTranspose2d(k4, s2, p1) forward and input gradient the way plan_base._deconv / _deconv_bwd lay them out: four 2x2 stride-1
    phase convolutions each (strided output view / strided input view, padding 0 | 1 per phase, accumulate + 0/1 mask on the partials).
    Returns (y NCHW, dx NCHW, kernel names of the launches)."""
    import ctypes as C
    from kurosiwo_amd import _lib, functional as Fk
    from kurosiwo_amd.runtime import DT as DTM, SrcSpec, make_conv, make_pack, packed_weight_numel, stream_ptr
    lib = _lib.load()
    dt = torch.bfloat16
    B, Cin, H, W = x.shape
    N = w.shape[1]
    xd = Fk.to_nhwc(x.to(dev), dt)
    wd = w.to(dev).contiguous()                       # [Cin][N][4][4]
    bd = bias.to(dev)
    out = torch.zeros((B, 2 * H, 2 * W, N), dtype=dt, device=dev)
    keep, names = [], []
    buf = C.create_string_buffer(4096)

    def launch(d):
        lib.ksmi_last_kernels(buf, 4096)
        _lib.check(lib.ksmi_conv_forward(C.byref(d), DTM[dt], stream_ptr()), "conv")
        if lib.ksmi_last_kernels(buf, 4096):
            names.append(buf.value.decode())

    def packed(wt, table, n_out, n_mod, sK, sN, tap_map):
        Npad = (n_out + 15) // 16 * 16
        o = torch.empty(packed_weight_numel(table, 4, Npad, dt), dtype=dt, device=dev)
        pd = make_pack(wt, o, table, 4, n_out, Npad, n_mod, sK, sN, 0, 1, 0, tap_map)
        _lib.check(lib.ksmi_pack_weights(C.byref(pd), DTM[dt], stream_ptr()), "pack")
        keep.extend([pd, o])
        return o
    for py in range(2):
        for px in range(2):
            tap_map = [(3 - 2 * a if py == 0 else 2 - 2 * a) * 4 + (3 - 2 * b if px == 0 else 2 - 2 * b) for a in range(2) for b in range(2)]
            d, table = make_conv([SrcSpec(xd, Cin)], [(out, N, 0, 0, N, 0)], out, bd, None, B, H, W, H, W, 2, 2, 1, 1 - py, N, dt, pad_x=1 - px,
                                 out_map=(2, 2, py, px, 2 * H, 2 * W))
            d.wpk = packed(wd, table, N, N, N * 16, 16, tap_map).data_ptr()
            keep.append(d)
            launch(d)
    y = Fk.to_nchw(out).cpu()
    dyd = Fk.to_nhwc(dy.to(dev), dt)
    dx = torch.zeros((B, H, W, Cin), dtype=dt, device=dev)
    md = Fk.to_nhwc(mask_t.to(dev), dt)
    zero, one = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    first = True
    for py in range(2):
        for px in range(2):
            tap_map = [(2 * a if py else 1 + 2 * a) * 4 + (2 * b if px else 1 + 2 * b) for a in range(2) for b in range(2)]
            d, table = make_conv([SrcSpec(dyd, N)], [(dx, Cin, 0, 0, Cin, 0 if first else 1)], dx, None, None, B, H, W, H, W, 2, 2, 1, py, Cin, dt,
                                 mask=(md, zero, one, one, zero), pad_x=px, in_map=(2, 2, py, px, 2 * H, 2 * W))
            d.wpk = packed(wd, table, Cin, Cin, 16, N * 16, tap_map).data_ptr()
            keep.append(d)
            launch(d)
            first = False
    # weight gradient: four 2x2 phase gradients over the parity sub-images of dy, each writing its 4 of the 16 taps (plan_base._deconv_wgrad)
    from kurosiwo_amd.runtime import make_wgrad
    gw = torch.zeros((Cin, N, 4, 4), dtype=torch.float32, device=dev)
    wnames = []
    for py in range(2):
        for px in range(2):
            tap_off = [(2 * a if py else 1 + 2 * a) * 4 + (2 * b if px else 1 + 2 * b) for a in range(2) for b in range(2)]
            dw, ws = make_wgrad([SrcSpec(dyd, N)], xd, Cin, 0, Cin, gw, 16, N * 16, 0, 0, B, H, W, H, W, 2, 2, 1, py, dt,
                                pad_x=px, in_map=(2, 2, py, px, 2 * H, 2 * W), tap_off=tap_off)
            scratch = torch.empty(max(int(ws), 256), dtype=torch.uint8, device=dev)
            dw.partial = scratch.data_ptr()
            keep.extend([dw, scratch])
            lib.ksmi_last_kernels(buf, 4096)
            _lib.check(lib.ksmi_conv_wgrad(C.byref(dw), DTM[dt], stream_ptr()), "wgrad")
            if lib.ksmi_last_kernels(buf, 4096):
                wnames.append(buf.value.decode())
    torch.cuda.synchronize()
    return y, Fk.to_nchw(dx).cpu(), names, gw.cpu(), wnames


@pytest.mark.parametrize("cfg", [dict(B=2, H=24, W=24, Cin=128, N=128), dict(B=1, H=19, W=33, Cin=128, N=256)])
def test_k4s2_deconv_phases_on_the_ring_kernel(dev, cfg):
    """models/changeformer.py:329-336 (ConvTranspose2d(k4, s2, p1) of the decoder) as plan_base._deconv / _deconv_bwd launch it:
    since round 5 the eight 2x2 phase convolutions run on igemm4's 2x2 instance (chunk-granular schedule, strided views).  Against
    torch's conv_transpose2d and its input gradient on the bf16-quantised operands; the ReLU mask of the input gradient is a 0/1 tensor."""
    B, H, W, Cin, N = cfg["B"], cfg["H"], cfg["W"], cfg["Cin"], cfg["N"]
    tag = f"k4s2.{B}{H}{W}{Cin}{N}"
    x = seeded_tensor(tag + ".x", (B, Cin, H, W))
    w = seeded_tensor(tag + ".w", (Cin, N, 4, 4)) * (2.0 / (Cin * 4)) ** 0.5
    bias = seeded_tensor(tag + ".b", (N,)) * 0.1
    dy = seeded_tensor(tag + ".dy", (B, N, 2 * H, 2 * W))
    mask_t = (seeded_tensor(tag + ".m", (B, Cin, H, W)) > 0).float()
    xq = q(x).requires_grad_(True)
    wq = q(w).requires_grad_(True)
    y_ref = F.conv_transpose2d(xq, wq, bias, stride=2, padding=1)
    y_ref.backward(q(dy))
    dx_ref = xq.grad * mask_t
    # (maps this small would stay on the tile kernel: the forced variant + a 3-workgroup grid put them on the persistent kernel with
    # several tiles per workgroup; the second pass is the tile kernel as the cross-check of the harness)
    for env, want in ((dict(KSMI_IGEMM4_VAR="4,4", KSMI_IGEMM4_CUS="3"), "igemm4_kernel<4, 4, false"), (dict(KSMI_IGEMM4_VAR=None, KSMI_IGEMM4_CUS=None), "igemm2_fwd_kernel")):
        with _Env(**env):
            y, dx, names, gw, wnames = _deconv_k4s2_via_phases(x, w, bias, dy, mask_t, dev)
        assert (y - y_ref.detach()).abs().max() < 2.5e-2 * y_ref.abs().max(), env
        assert (dx - dx_ref).abs().max() < 3e-2 * dx_ref.abs().max(), env
        # weight gradient: fp32 accumulation of exact bf16 products on both sides
        assert (gw - wq.grad).abs().max() < 2e-3 * wq.grad.abs().max(), (env, float((gw - wq.grad).abs().max()), float(wq.grad.abs().max()))
        # ... on the channel-owner kernel's 2x2-window instances (round 5), whatever the convolution switches say
        assert len(wnames) == 4 and all("wgrad3_kernel<4, 1, 4, false, true, " in n for n in wnames), wnames
        assert len(names) == 8 and all(want in n for n in names), names
        if "igemm4" in want:
            assert all(n.rstrip(">").endswith("2, 2") for n in names), names
