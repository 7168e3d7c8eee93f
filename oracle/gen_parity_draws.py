"""Golden vectors for the per-draw K-step parity gate: the protocol of oracle/gen_parity_run.py --reference
(imported /root/reference SNUNet_ECAM + BCEandDiceLoss + torch.optim.Adam, 40 steps, batches of 4, held-out mIoU after 20 and 40
steps) repeated under the eleven weight perturbations tools/parity_draws.py uses on the GPU (conv0_0.conv1.weight scaled by 1 + p)
and twelve more,

  * in fp32                                     -> how far a 1e-7 perturbation alone moves the CPU fp32 run, and
  * with bf16 STORAGE emulated (oracle/bf16_storage.py: activations / stored gradients / MFMA weight operands rounded to bf16,
    fp32 accumulation, statistics, head, loss and optimiser)  -> what the arithmetic contract of the HIP performance mode does
    to the REFERENCE's own module graph.

TEST INFRASTRUCTURE, build container only.  Writes tests/golden/snunet_parity_draws_ref.npz: for every (mode, perturbation) the loss
trajectory, mIoU / per-class IoU at both checkpoints.  Runs are independent processes (DRAW_PROCS at a time, DRAW_THREADS threads each);
finished draws are cached under /tmp/parity_draws so the script can be resumed.

    python oracle/gen_parity_draws.py            # ~3 h on 8 cores (27 runs of ~25 minutes, four at a time)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the eleven perturbations of tools/parity_draws.py / tests/test_gpu_parity_gate.py first, then twelve more for the tail of the distribution
PERTURBS = (0.0, 1e-7, -1e-7, 2e-7, -2e-7, 3e-7, -3e-7, 5e-7, -5e-7, 1e-6, -1e-6,
            4e-7, -4e-7, 6e-7, -6e-7, 7e-7, -7e-7, 8e-7, -8e-7, 9e-7, -9e-7, 1.5e-6, -1.5e-6)
MODES = ("fp32", "bf16emu")
NDRAWS = {"fp32": 4, "bf16emu": len(PERTURBS)}      # fp32 moves by < 5e-4 under these perturbations: four draws show it
CACHE = os.environ.get("DRAW_CACHE", "/tmp/parity_draws")


def one(mode, pz, path):
    import torch
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from models.snunet import SNUNet_ECAM                      # (reference)
    from utilities.bce_and_dice import BCEandDiceLoss          # (reference)
    from oracle import bf16_storage, metrics_ref
    from oracle.gen_parity_run import BATCH, CHECKPOINTS, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
    from oracle.seeded import seeded_fill_
    torch.set_num_threads(int(os.environ.get("DRAW_THREADS", "2")))
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()
    model = SNUNet_ECAM(2, 3, base_channel=32)
    seeded_fill_(model.state_dict())
    if pz:
        with torch.no_grad():
            model.conv0_0.conv1.weight.mul_(1.0 + pz)
    if mode == "bf16emu":
        bf16_storage.attach(model)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    criterion = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    losses, out = [], {}
    model.train()
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        optimizer.zero_grad()
        loss = criterion(model(xA[s:s + BATCH], xB[s:s + BATCH]), mask[s:s + BATCH])
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        if k + 1 in CHECKPOINTS:
            cm = np.zeros((4, 4), np.int64)
            model.eval()
            with torch.no_grad():
                for e in range(0, HELD_OUT, 8):
                    logits = model(eA[e:e + 8], eB[e:e + 8])
                    cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits.numpy()), emask[e:e + 8].numpy())
            model.train()
            m = metrics_ref.metrics_from_cm(cm)
            out[f"miou{k + 1}"], out[f"iou{k + 1}"] = np.array(m["miou"]), m["iou"]
            print(f"{mode} {pz:+g} K={k + 1}: mIoU {float(m['miou']):.5f}", flush=True)
    np.savez(path, losses=np.array(losses), **out)


def main():
    os.makedirs(CACHE, exist_ok=True)
    jobs = [(m, p, os.path.join(CACHE, f"{m}_{i}.npz")) for m in reversed(MODES) for i, p in enumerate(PERTURBS[:NDRAWS[m]])]
    todo = [j for j in jobs if not os.path.exists(j[2])]
    procs, nproc = [], int(os.environ.get("DRAW_PROCS", "4"))
    while todo or procs:
        procs = [p for p in procs if p.poll() is None]
        while todo and len(procs) < nproc:
            m, p, path = todo.pop(0)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", m, repr(p), path]))
        if procs:
            procs[0].wait()
    res = {"perturbations": np.array(PERTURBS)}
    for m in MODES:
        runs = [np.load(os.path.join(CACHE, f"{m}_{i}.npz")) for i in range(NDRAWS[m])]
        res[f"{m}.losses"] = np.stack([r["losses"] for r in runs])
        for k in (20, 40):
            res[f"{m}.miou{k}"] = np.array([float(r[f"miou{k}"]) for r in runs])
            res[f"{m}.iou{k}"] = np.stack([r[f"iou{k}"] for r in runs])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "snunet_parity_draws_ref.npz"), **res)
    for m in MODES:
        print(m, "K=20", np.round(res[f"{m}.miou20"], 5).tolist(), "\n   K=40", np.round(res[f"{m}.miou40"], 5).tolist())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2], float(sys.argv[3]), sys.argv[4])
    else:
        main()
