"""GPU parity of the Unet(resnet18) path (row U1; kurosiwo_amd/unet.py) against the CPU restatement oracle/unet_ref.py.
PARITY UNPINNED: the reference's model comes from segmentation-models-pytorch 0.3.2, absent from /root/reference and from this image;
the oracle restates its published architecture (see its header), so these tests pin the HIP kernels to that restatement only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]


def sar_like(name, shape):
    from oracle.seeded import seeded_tensor
    return seeded_tensor(name, shape).clamp_(-2.23, 5.75)


def build(precision):
    from kurosiwo_amd.unet import Unet
    from oracle import unet_ref as U
    from oracle.seeded import seeded_fill_
    model = Unet("resnet18", encoder_weights=None, in_channels=2, classes=3, precision=precision)
    sd = seeded_fill_(U.new_state_dict(2, 3))
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    return model.cuda(), sd


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def test_eval_forward():
    from oracle import unet_ref as U
    model, sd = build("fp32")
    model.eval()
    x = sar_like("unet.eval.x", (1, 2, 224, 224))
    with torch.no_grad():
        ref = U.unet_forward(sd, x, training=False)
        out = model(x.cuda())
    assert relerr(out.cpu(), ref) < 1e-3
    margin = ref.topk(2, dim=1).values
    confident = (margin[:, 0] - margin[:, 1]) > 1e-3 * float(ref.abs().max())
    assert (out.argmax(1).cpu() == ref.argmax(1))[confident].all()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_forward_backward(precision):
    from oracle import unet_ref as U
    from oracle.seeded import seeded_labels
    B = 2
    model, sd = build(precision)
    model.train()
    x = sar_like("unet.train.x", (B, 2, 224, 224))
    lbl = seeded_labels("unet.train.lbl", (B, 224, 224))
    logits = model(x.cuda())
    plan = model.plan(B, 224, 224, True, True)
    inter = {}
    with torch.no_grad():
        ref = U.unet_forward(sd, x, training=True, inter=inter)
    tol = 1e-3 if precision == "fp32" else 0.15
    errs = {f"f{i}": relerr(nchw(plan.named[f"f{i}"]), inter[f"f{i}"]) for i in range(1, 6)}
    errs.update({f"d{i}": relerr(nchw(plan.named[f"d{i}"]), inter[f"d{i}"]) for i in range(5)})
    errs["logits"] = relerr(logits.detach().cpu(), ref)
    assert not {k: v for k, v in errs.items() if not v < tol}, errs
    loss = torch.nn.functional.cross_entropy(logits, lbl.cuda(), weight=torch.tensor(CLASS_WEIGHTS, device="cuda"), ignore_index=3)
    loss.backward()
    # the oracle backward runs on the GPU's ReLU active sets (materialised ReLU outputs: stem, block outputs, decoder outputs; the
    # operand-fused ReLUs are recomputed from the GPU's pre-activations and BatchNorm scale/shift)
    _, ref_loss, ref_grads, _ = U.loss_and_grads(sd, x, lbl, CLASS_WEIGHTS)
    assert abs(float(loss) - ref_loss) < (1e-3 if precision == "fp32" else 5e-2)
    coss, worst = [], {}
    for k, p in model.named_parameters():
        g, r = p.grad.detach().float().cpu(), ref_grads[k]
        if float(r.abs().max()) < 1e-12:
            continue
        cos = float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm() + 1e-30))
        coss.append(cos)
        l2 = float((g - r).double().norm() / (r.double().norm() + 1e-30))
        if precision == "fp32" and not l2 < 3e-2:      # unmasked oracle: isolated ReLU flips perturb single gradients (see test_gpu_floodvit)
            worst[k] = l2
        if precision == "bf16" and not cos > 0.3:      # sanity only: 20 BatchNorm layers down to 98 samples amplify the bf16 rounding of this random net
            worst[k] = cos
    assert float(np.median(coss)) > (0.9999 if precision == "fp32" else 0.75), float(np.median(coss))
    assert not worst, f"{precision}: {len(worst)}: {dict(list(worst.items())[:10])}"


def test_main_entry_unet_end_to_end_tiny(tmp_path, monkeypatch):
    import shutil
    import main as entry
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copytree(os.path.join(root, "configs"), tmp_path / "configs")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("KSMI_SYNTHETIC_TILES", "8,4,4")
    miou = entry.main(["--method", "unet", "--inputs", "post_event", "--batch_size", "4"])
    assert 0.0 <= miou <= 100.0
