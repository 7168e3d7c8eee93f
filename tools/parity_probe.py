"""K-step parity protocol of tests/test_gpu_parity_gate.py with a relative perturbation of one weight tensor (PERTURB=...): how far do the
checkpoints move with the rounding realisation?  usage: PERTURB=1e-7 python tools/parity_probe.py bf16|fp32"""
import os, sys
root = os.getcwd(); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.trainer import CDTrainStep
from oracle import metrics_ref, snunet_ref as R
from oracle.gen_parity_run import BATCH, CHECKPOINTS, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
from oracle.seeded import seeded_fill_
precision = sys.argv[1]
gold = np.load(os.path.join(root, "tests", "golden", "snunet_parity_run_ref.npz"))
dev = torch.device("cuda:0")
(xA, xB, mask), (eA, eB, emask) = protocol_tiles()
sd = seeded_fill_(R.new_state_dict(2, 3, 32))
pz = float(os.environ.get("PERTURB", "0"))
if pz:
    sd["conv0_0.conv1.weight"] = sd["conv0_0.conv1.weight"] * (1.0 + pz)
model = SNUNet_ECAM(2, 3, base_channel=32, precision=precision)
model.load_state_dict(sd)
model = model.to(dev).train()
step = CDTrainStep(model, BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3)
def evaluate(m):
    m.eval(); cm = np.zeros((4, 4), np.int64)
    with torch.no_grad():
        for s in range(0, HELD_OUT, 8):
            logits = m(eA[s:s + 8].to(dev), eB[s:s + 8].to(dev)).float().cpu().numpy()
            cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits), emask[s:s + 8].numpy())
    m.train(); return metrics_ref.metrics_from_cm(cm)
losses = []
for k in range(K_STEPS):
    s = (k % (TRAIN_TILES // BATCH)) * BATCH
    losses.append(float(step.step(xA[s:s + BATCH].to(dev), xB[s:s + BATCH].to(dev), mask[s:s + BATCH].to(dev))[0]))
    if k + 1 in CHECKPOINTS:
        m = evaluate(model)
        print(f"{precision} perturb {pz:g} K={k + 1}: mIoU {m['miou']:.5f} delta {float(m['miou']) - float(gold[f'miou{k + 1}']):+.5f} loss {losses[-1]:.5f} vs {gold['losses'][k]:.5f}", flush=True)
rel = np.abs(np.array(losses) - gold["losses"]) / gold["losses"]
print(f"{precision} perturb {pz:g} max rel loss deviation {rel.max():.3f} at step {int(rel.argmax())}; losses 15..22: {np.round(losses[15:23], 4).tolist()}")
