"""tests/golden/snunet_dem_shard_bf16emu.npz: the train step of tests/test_gpu_bench_size.py::test_snunet_dem_shard_vs_reference_golden
(BASELINE.json configs[2] per-GPU shard: SNUNet-ECAM c = 3, batch 8, 224 x 224) on the IMPORTED reference with bf16 STORAGE emulated
(oracle/bf16_storage.py): what the reference's own module graph computes under the arithmetic contract of the HIP performance mode.
The bf16 HIP path is held to THESE vectors (the fp32 ones of snunet_dem_shard.npz bound it only loosely: bf16 storage moves the
first block's gradient norms by -5 ... -9 % on the reference itself, LABNOTES.md round 5).

TEST INFRASTRUCTURE, build container only (imports /root/reference).      python oracle/gen_bf16emu_golden.py      (~1 minute)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from kurosiwo_amd.synthetic import cd_inputs, make_batch  # noqa: E402
from oracle import bf16_storage  # noqa: E402
from oracle.seeded import seeded_fill_, seeded_tensor  # noqa: E402


def main():
    from models.snunet import SNUNet_ECAM                      # (reference)
    from utilities.bce_and_dice import BCEandDiceLoss          # (reference)
    torch.set_num_threads(8)
    B = 8
    (xA, xB), lbl = cd_inputs(make_batch(B, 224, 224, seed=4321), ("pre_event_1", "post_event"))
    dem = torch.nn.functional.interpolate(seeded_tensor("snunet_dem_shard.dem", (B, 1, 14, 14)), size=(224, 224), mode="bilinear", align_corners=False)
    out = {}
    for mode in ("fp32", "bf16emu"):
        model = SNUNet_ECAM(3, 3, base_channel=32)
        seeded_fill_(model.state_dict())
        if mode == "bf16emu":
            bf16_storage.attach(model)
        model.train()
        logits = model(torch.cat((xA, dem), 1), torch.cat((xB, dem), 1))
        loss = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)(logits, lbl)
        loss.backward()
        out[mode] = (float(loss.detach()), logits.detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
        print(mode, "loss", out[mode][0], flush=True)
    l32, lg32, g32 = out["fp32"]
    le, lge, ge = out["bf16emu"]
    res = {"train_loss": np.array(le), "train_loss_fp32": np.array(l32), "train_logits_sub": lge[:, :, ::8, ::8].numpy(),
           "train_logits_absmax": np.array(float(lg32.abs().max()))}
    for k in ge:
        cos = float((ge[k] * g32[k]).sum() / (ge[k].norm() * g32[k].norm() + 1e-30))
        res[f"gstat.{k}"] = np.array([float(ge[k].double().norm()), float(g32[k].double().norm()), cos])
    for k in ("conv0_0.conv1.weight", "conv0_0.conv2.weight", "conv0_4.conv2.weight", "conv_final.weight"):
        res[f"grad.{k}"] = ge[k].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "snunet_dem_shard_bf16emu.npz"), **res)
    worst = sorted(((res[f"gstat.{k}"][0] / max(res[f"gstat.{k}"][1], 1e-30), k) for k in ge if not k.endswith("conv2.bias")))
    print("norm ratio emulated / fp32: lowest", worst[:5], "highest", worst[-3:])


if __name__ == "__main__":
    main()
