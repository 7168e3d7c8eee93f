"""per-parameter gradient cosine of the bf16 SNUNet step against the CPU fp32 oracle (tests/test_gpu_snunet.py bf16 case), for A/B of kernels"""
import os, sys
root = os.getcwd()
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, numpy as np
import oracle.snunet_ref as R
from oracle.seeded import seeded_fill_, seeded_labels, seeded_tensor
CLASS_WEIGHTS = [0.3715753140309927, 14.009780283125977, 8.20405370357821]
def sar_like(name, shape): return seeded_tensor(name, shape).clamp_(-2.23, 5.75)
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.loss import BCEandDiceLoss
c, bc, B, H, W = 2, 32, 2, 64, 64
tag = f"bf{c}{bc}{B}{H}{W}"
xA, xB = sar_like(tag + "A", (B, c, H, W)), sar_like(tag + "B", (B, c, H, W))
lbl = seeded_labels(tag + "L", (B, H, W))
sd = seeded_fill_(R.new_state_dict(c, 3, bc))
m = SNUNet_ECAM(c, 3, base_channel=bc, precision="bf16"); m.load_state_dict({k: v.clone() for k, v in sd.items()}); m = m.cuda().train()
logits = m(xA.cuda(), xB.cuda())
loss = BCEandDiceLoss(CLASS_WEIGHTS, 3, True)(logits, lbl.cuda()); loss.backward()
ref_loss, ref_logits, ref_grads = R.train_step(sd, R.AdamRef(sd, lr=0.0), xA, xB, lbl, CLASS_WEIGHTS, True)
print("logits rel", float((logits.detach().cpu() - ref_logits).abs().max() / ref_logits.abs().max()))
out = {}
for k, p in m.named_parameters():
    g, r = p.grad.cpu().flatten().double(), ref_grads[k].flatten().double()
    if float(r.norm()) > 1e-6:
        out[k] = float((g @ r) / (g.norm() * r.norm() + 1e-30))
print("median", np.median(list(out.values())))
for k, v in out.items(): print(f"{v:.4f} {k}")
