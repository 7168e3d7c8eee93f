"""the Dataset's per-tile pipeline (dataset/Dataset.py:164-168 clamp to [0, clamp_input] + nan_to_num(clamp_input);
:193-198 Normalize) folded into the image load of SNUNet's first convolution (ksmi_conv_first_forward_raw / ksmi_im2col3x3_raw).

Oracle: the reference's own torch expressions on the CPU (clamp -> nan_to_num -> Normalize), then the unfused path.  Integer-free
but elementwise fp32 with the same operations in the same order, so the bar is bit-exact: logits AND every gradient."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN, STD, CLAMP = (0.0953, 0.0264), (0.0427, 0.0215), 0.15


def _raw_tiles(B, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.distributions.Gamma(torch.tensor(4.0), torch.tensor(4.0)).sample((B, C, H, W)) * 0.09
    u = torch.rand((B, C, H, W), generator=g)
    x[u < 0.02] = float("nan")                 # no-data pixels of the archive
    x[(u > 0.02) & (u < 0.03)] = -0.01         # below the clamp
    x[(u > 0.03) & (u < 0.05)] = 3.0           # above it
    x[(u > 0.05) & (u < 0.055)] = float("inf")
    return x.float()


def _reference_pipeline(x, mean, std, clamp):
    """dataset/Dataset.py:164-168 + :193-198 with torch on the CPU"""
    y = torch.clamp(x, min=0.0, max=clamp)
    y = torch.nan_to_num(y, clamp)
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    return (y - m) / s


def _model(cin, precision, seed=3):
    from kurosiwo_amd.snunet import SNUNet_ECAM
    torch.manual_seed(seed)
    return SNUNet_ECAM(cin, 3, base_channel=16, precision=precision).cuda().train()


def _step(model, xA, xB, lbl):
    for p in model.parameters():
        p.grad = None
    out = model(xA, xB)
    loss = torch.nn.functional.cross_entropy(out, lbl, ignore_index=3)
    loss.backward()
    return out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_raw_tiles_through_the_first_conv_equal_preprocess_then_forward(precision):
    from kurosiwo_amd.data import preprocess_gpu
    B, C, H, W = 2, 2, 64, 48
    rA, rB = _raw_tiles(B, C, H, W, 11), _raw_tiles(B, C, H, W, 12)
    lbl = torch.randint(0, 4, (B, H, W), generator=torch.Generator().manual_seed(5)).cuda()
    # the standalone GPU preprocess equals the reference's CPU expressions bit for bit
    nA = preprocess_gpu(rA.cuda(), MEAN, STD, CLAMP)
    assert torch.equal(nA.cpu(), _reference_pipeline(rA, MEAN, STD, CLAMP))
    nB = preprocess_gpu(rB.cuda(), MEAN, STD, CLAMP)
    model = _model(C, precision)
    out0, g0 = _step(model, nA, nB, lbl)
    model.set_input_pipeline(MEAN, STD, CLAMP)
    out1, g1 = _step(model, rA.cuda(), rB.cuda(), lbl)
    assert torch.isfinite(out1).all()
    assert torch.equal(out0, out1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # back to normalised inputs: the plans of both modes coexist
    model.set_input_pipeline()
    out2, _ = _step(model, nA, nB, lbl)
    assert torch.equal(out0, out2)


def test_raw_tiles_with_a_dem_channel_that_is_not_clamped():
    """configs['dem']: the DEM rides as a third input channel (change_detection_trainer.py:117-133); it is normalised with
    dem_mean / dem_std and never clamped (Dataset.py:741-779) -> clamp[c] < 0"""
    B, H, W = 2, 32, 32
    mean, std, clamp = MEAN + (93.4313,), STD + (1410.8382,), (CLAMP, CLAMP, -1.0)
    rA, rB = _raw_tiles(B, 3, H, W, 21), _raw_tiles(B, 3, H, W, 22)
    for r in (rA, rB):
        dem = 200 + 150 * torch.sin(torch.linspace(0, 6, H)).view(1, H, 1) * torch.ones(B, H, W)
        dem[:, 3, 4] = float("nan")
        dem[:, 7, :] = -40.0                                       # below sea level: must NOT be clamped to 0
        r[:, 2] = dem
    lbl = torch.randint(0, 3, (B, H, W), generator=torch.Generator().manual_seed(6)).cuda()

    def ref(r):
        sar = _reference_pipeline(r[:, :2], MEAN, STD, CLAMP)
        d = torch.nan_to_num(r[:, 2:3], nan=mean[2])               # rioxarray fills the gaps before Normalize; here: the mean (-> 0)
        return torch.cat((sar, (d - torch.tensor(mean[2])) / torch.tensor(std[2])), 1)
    model = _model(3, "fp32")
    out0, g0 = _step(model, ref(rA).cuda(), ref(rB).cuda(), lbl)
    model.set_input_pipeline(mean, std, clamp)
    out1, g1 = _step(model, rA.cuda(), rB.cuda(), lbl)
    assert torch.equal(out0, out1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # the DEM as a shared tail: forward(xA, xB, dem) == forward(cat(xA, dem), cat(xB, dem)), no concatenated copies
    demA = rA[:, 2:3].clone()
    rB[:, 2:3] = demA
    out2, g2 = _step(model, rA.cuda(), rB.cuda(), lbl)

    def step_tail():
        for p in model.parameters():
            p.grad = None
        out = model(rA[:, :2].cuda(), rB[:, :2].cuda(), demA.cuda())
        torch.nn.functional.cross_entropy(out, lbl, ignore_index=3).backward()
        return out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    out3, g3 = step_tail()
    assert torch.equal(out2, out3)
    for k in g2:
        assert torch.equal(g2[k], g3[k]), k
    with pytest.raises(ValueError):
        model(rA[:, :2].cuda(), rB[:, :2].cuda())                  # a channel short
    with pytest.raises(ValueError):
        model(rA.cuda(), rB.cuda(), demA.cuda())                   # a channel too many


def test_argument_errors_are_loud():
    from kurosiwo_amd import _lib
    from kurosiwo_amd.runtime import DT
    model = _model(2, "fp32")
    with pytest.raises(ValueError):
        model.set_input_pipeline(MEAN + (1.0,), STD + (1.0,), CLAMP)
    with pytest.raises(ValueError):
        model.set_input_pipeline(MEAN, (0.0, 1.0), CLAMP)
    lib = _lib.load()
    x = torch.zeros(1, 2, 16, 16, device="cuda")
    w = torch.zeros(32, 2, 3, 3, device="cuda")
    b = torch.zeros(32, device="cuda")
    out = torch.zeros(1, 16, 16, 32, device="cuda")
    m = torch.zeros(2, device="cuda")
    rc = lib.ksmi_conv_first_forward_raw(x.data_ptr(), None, 2, w.data_ptr(), b.data_ptr(), out.data_ptr(), None, 1, 2, 16, 16, 32,
                                         m.data_ptr(), None, None, DT[torch.float32], None)
    assert rc != 0 and b"come together" in lib.ksmi_last_error()
    rc = lib.ksmi_conv_first_forward_raw(x.data_ptr(), x.data_ptr(), 2, w.data_ptr(), b.data_ptr(), out.data_ptr(), None, 1, 2, 16, 16, 32,
                                         None, None, None, DT[torch.float32], None)
    assert rc != 0 and b"c_head" in lib.ksmi_last_error()
