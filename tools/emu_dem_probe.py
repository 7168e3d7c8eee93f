"""CPU (build container): first-block gradient norms of the imported reference, fp32 vs bf16-storage emulation (oracle/bf16_storage.py),
same inputs as tests/test_gpu_bench_size.py::test_snunet_dem_shard_vs_reference_golden.  PERTURB = relative change of conv0_0.conv1.weight."""
import os, sys
root = os.getcwd(); sys.path.insert(0, root); sys.path.insert(0, "/root/reference"); sys.dont_write_bytecode = True
import numpy as np, torch
from kurosiwo_amd.synthetic import cd_inputs, make_batch
from oracle import snunet_ref as R, bf16_storage
from oracle.seeded import seeded_fill_, seeded_tensor
from models.snunet import SNUNet_ECAM
from utilities.bce_and_dice import BCEandDiceLoss
torch.set_num_threads(int(os.environ.get("THREADS", "8")))
gold = np.load(os.path.join(root, "tests", "golden", "snunet_dem_shard.npz"))
B = 8
(xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
keys = ("conv0_0.conv1.weight", "conv0_0.bn1.weight", "conv0_0.bn1.bias", "conv0_0.conv2.weight", "conv0_0.bn2.weight", "conv0_1.conv1.weight", "conv1_0.conv1.weight", "conv0_4.conv2.weight")
ref_grads = None
for mode in os.environ.get("MODES", "fp32,emu").split(","):
    for pz in [float(v) for v in os.environ.get("PERTURB", "0").split(",")]:
        model = SNUNet_ECAM(3, 3, base_channel=32)
        sd = seeded_fill_(model.state_dict())
        if pz:
            with torch.no_grad():
                model.conv0_0.conv1.weight.mul_(1.0 + pz)
        if mode.startswith("emu"):
            import re
            what = tuple(os.environ.get("WHAT", "w,conv,relu,up").split(","))
            pat = os.environ.get("ONLY")
            bf16_storage.attach(model, round_grads=(mode != "emu_fwd"), round_inputs=os.environ.get("ROUND_IN", "1") == "1", what=what,
                                only=(lambda n: re.search(pat, n) is not None) if pat else None)
        model.train()
        loss = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)(model(torch.cat((xA, dem), 1), torch.cat((xB, dem), 1)), lbl)
        loss.backward()
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        if ref_grads is None:
            ref_grads = g
        print(f"{mode} perturb {pz:+g}: loss {float(loss):.6f} (golden {float(gold['train_loss']):.6f})")
        for k in keys:
            cos = float((g[k] * ref_grads[k]).sum() / (g[k].norm() * ref_grads[k].norm()))
            print(f"   {k:28s} {float(g[k].double().norm()):.4f}  golden {float(gold[f'gstat.{k}'][0]):.4f}  cos vs first {cos:.4f}", flush=True)
