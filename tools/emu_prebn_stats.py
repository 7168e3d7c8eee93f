"""CPU: per-channel |mean| / std of every pre-BatchNorm tensor of the imported reference SNUNet on the dem-shard inputs, and the bf16
quantisation noise of that tensor relative to its per-channel std (the quantity the BatchNorm backward scale 1/sigma sees)."""
import os, sys
root = os.getcwd(); sys.path.insert(0, root); sys.path.insert(0, "/root/reference"); sys.dont_write_bytecode = True
import numpy as np, torch
from kurosiwo_amd.synthetic import cd_inputs, make_batch
from oracle import bf16_storage
from oracle.seeded import seeded_fill_, seeded_tensor
from models.snunet import SNUNet_ECAM
torch.set_num_threads(8)
B = 8
(xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
model = SNUNet_ECAM(3, 3, base_channel=32); seeded_fill_(model.state_dict()); model.train()
rows = []
def hook(name):
    def h(m, inp, out):
        x = inp[0].detach()
        mu = x.mean((0, 2, 3)); sd = x.std((0, 2, 3))
        q = (bf16_storage.bf16_round(x) - x)
        qs = q.pow(2).mean((0, 2, 3)).sqrt()
        rows.append((name, float((mu.abs() / sd).median()), float((mu.abs() / sd).max()), float((qs / sd).median()), float((qs / sd).max())))
    return h
for n, m in model.named_modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.register_forward_hook(hook(n))
with torch.no_grad():
    model(torch.cat((xA, dem), 1), torch.cat((xB, dem), 1))
print(f"{'bn':16s} med|mu|/sd  max|mu|/sd  med q/sd   max q/sd")
seen = set()
for r in rows:
    print(f"{r[0]:16s} {r[1]:9.3f} {r[2]:10.3f} {r[3]:10.5f} {r[4]:10.5f}")
