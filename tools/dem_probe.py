import os, sys
root = os.getcwd(); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from kurosiwo_amd.loss import BCEandDiceLoss
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.synthetic import cd_inputs, make_batch
from oracle import snunet_ref as R
from oracle.seeded import seeded_fill_, seeded_tensor
gold = np.load(os.path.join(root, "tests", "golden", "snunet_dem_shard.npz"))
B = 8
(xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
sd = seeded_fill_(R.new_state_dict(3, 3, 32))
if os.environ.get("PERTURB"):
    sd["conv0_0.conv1.weight"] = sd["conv0_0.conv1.weight"] * (1.0 + float(os.environ["PERTURB"]))
for prec in ("bf16",):
    m = SNUNet_ECAM(3, 3, base_channel=32, precision=prec); m.load_state_dict({k: v.clone() for k, v in sd.items()}); m = m.cuda().train()
    logits = m(xA.cuda(), xB.cuda(), dem.cuda())
    loss = BCEandDiceLoss([1.0, 1.0, 1.0], 3, True)(logits, lbl.cuda()); loss.backward()
    out = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
    for k in ("conv0_0.conv1.weight", "conv0_0.bn1.weight", "conv0_0.bn1.bias", "conv0_0.conv2.weight", "conv0_0.bn2.weight", "conv0_1.conv1.weight", "conv1_0.conv1.weight", "conv0_4.conv2.weight"):
        print(prec, k, float(out[k].double().norm()), float(gold[f"gstat.{k}"][0]))
