"""Distribution of the K-step parity protocol (tests/test_gpu_parity_gate.py) over rounding realisations: one HIP run per relative
perturbation of conv0_0.conv1.weight, optionally on another build of the library (KSMI_LIB=<path to a libksmi.so>).
usage: [KSMI_LIB=...] python tools/parity_draws.py bf16|fp32 [perturbation ...]"""
import os, sys
root = os.getcwd(); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import kurosiwo_amd._lib as L
if os.environ.get("KSMI_LIB"):
    L.LIB_PATH = os.environ["KSMI_LIB"]
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.trainer import CDTrainStep
from oracle import metrics_ref, snunet_ref as R
from oracle.gen_parity_run import BATCH, CHECKPOINTS, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
from oracle.seeded import seeded_fill_
precision = sys.argv[1]
draws = [float(a) for a in sys.argv[2:]] or [0.0, 1e-7, -1e-7, 2e-7, -2e-7, 3e-7, -3e-7, 5e-7, -5e-7, 1e-6, -1e-6]
gold = np.load(os.path.join(root, "tests", "golden", "snunet_parity_run_ref.npz"))
dev = torch.device("cuda:0")
(xA, xB, mask), (eA, eB, emask) = protocol_tiles()
def evaluate(m):
    m.eval(); cm = np.zeros((4, 4), np.int64)
    with torch.no_grad():
        for s in range(0, HELD_OUT, 8):
            logits = m(eA[s:s + 8].to(dev), eB[s:s + 8].to(dev)).float().cpu().numpy()
            cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits), emask[s:s + 8].numpy())
    m.train(); return metrics_ref.metrics_from_cm(cm)
print(f"library {L.LIB_PATH}; {precision}; env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("KSMI_")))
rows = []
for pz in draws:
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    if pz:
        sd["conv0_0.conv1.weight"] = sd["conv0_0.conv1.weight"] * (1.0 + pz)
    model = SNUNet_ECAM(2, 3, base_channel=32, precision=precision)
    model.load_state_dict(sd)
    model = model.to(dev).train()
    step = CDTrainStep(model, BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3)
    losses, d = [], {}
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        losses.append(float(step.step(xA[s:s + BATCH].to(dev), xB[s:s + BATCH].to(dev), mask[s:s + BATCH].to(dev))[0]))
        if k + 1 in CHECKPOINTS:
            d[k + 1] = float(evaluate(model)["miou"]) - float(gold[f"miou{k + 1}"])
    rel = np.abs(np.array(losses) - gold["losses"]) / gold["losses"]
    rows.append((pz, d[20], d[40], rel.max(), int(rel.argmax())))
    all_losses = globals().setdefault("all_losses", [])
    all_losses.append(losses)
    print(f"  perturb {pz:+.0e}: K=20 {d[20]:+.5f}  K=40 {d[40]:+.5f}  max rel loss dev {rel.max():.2f} at step {int(rel.argmax())}", flush=True)
a = np.array([[r[1], r[2]] for r in rows])
print(f"  K=20: median {np.median(a[:, 0]):+.5f}  min {a[:, 0].min():+.5f}  max {a[:, 0].max():+.5f}   |   K=40: median {np.median(a[:, 1]):+.5f}  min {a[:, 1].min():+.5f}  max {a[:, 1].max():+.5f}")

if os.environ.get("DRAWS_OUT"):
    np.savez(os.environ["DRAWS_OUT"], perturbations=np.array(draws), losses=np.array(all_losses), d20=a[:, 0], d40=a[:, 1])
