"""Golden vectors of the bf16 parity gate of SURVEY.md §8(d)(ii): "mIoU within +-0.002 of CPU fp32 after an identical K-step training
run from identical weights on the 64 held-out tiles (seed 424242)".

TEST INFRASTRUCTURE.  Runs the CPU oracle (oracle/snunet_ref.py: pinned to /root/reference/models/snunet.py + utilities/bce_and_dice.py
by tests/golden/snunet_*.npz) -- fp32, 40 Adam steps of ce+dice on batches of 4 synthetic tiles, then eval-mode inference on the 64
held-out tiles after 20 and after 40 steps -- and stores the loss trajectory, the 4x4 confusion matrix, per-class IoU and mIoU in tests/golden/snunet_parity_run.npz.
The GPU test (tests/test_gpu_parity_gate.py) repeats the run on the HIP path in bf16 and in fp32 from the same weights and tiles.

    python oracle/gen_parity_run.py              # the oracle's run, ~3 minutes on 8 CPU threads
    python oracle/gen_parity_run.py --reference  # the SAME protocol on the imported reference (/root/reference/models/snunet.py,
                                                 # utilities/bce_and_dice.py, torch.optim.Adam as change_detection_trainer.py:45-66
                                                 # builds it): tests/golden/snunet_parity_run_ref.npz -- the vectors the gate uses
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kurosiwo_amd.synthetic import cd_inputs, make_batch  # noqa: E402  (host-side tile generator, no device code)
from oracle import metrics_ref, snunet_ref as R  # noqa: E402
from oracle.seeded import seeded_fill_  # noqa: E402

K_STEPS, TRAIN_TILES, BATCH, HELD_OUT, HELD_OUT_SEED, TRAIN_SEED = 40, 8, 4, 64, 424242, 31337
CHECKPOINTS = (20, 40)      # evaluate after this many steps: 20 = still on the steep part of the learning curve, 40 = on the plateau
# PLATEAU protocol (round 6): the same run where the protocol is not the noise source -- batches of 16 (half the
# benchmarked per-GPU batch: what the fp32 reference fits into the build container's 62 GB), 80 Adam steps, evaluated after 60 and 80
# steps (both far onto the plateau).  tests/test_gpu_parity_gate.py asserts EVERY bf16 draw within the survey's +-0.002 there.
P_K_STEPS, P_TRAIN_TILES, P_BATCH, P_CHECKPOINTS = 80, 32, 16, (60, 80)


def protocol_tiles():
    (xA, xB), mask = cd_inputs(make_batch(TRAIN_TILES, seed=TRAIN_SEED), ("pre_event_1", "post_event"))
    (eA, eB), emask = cd_inputs(make_batch(HELD_OUT, seed=HELD_OUT_SEED), ("pre_event_1", "post_event"))
    return (xA, xB, mask), (eA, eB, emask)


def main():
    torch.set_num_threads(8)
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()
    sd = seeded_fill_(R.new_state_dict(2, 3, 32))
    opt = R.AdamRef(sd, lr=1e-3)
    losses, out = [], {}

    def evaluate(tag):
        cm = np.zeros((4, 4), np.int64)
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                logits = R.snunet_forward(sd, eA[s:s + 8], eB[s:s + 8], training=False)
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits.numpy()), emask[s:s + 8].numpy())
        m = metrics_ref.metrics_from_cm(cm)
        print(tag, "cm\n", cm, "\niou", m["iou"], "miou", m["miou"], flush=True)
        out[f"cm{tag}"], out[f"iou{tag}"], out[f"miou{tag}"], out[f"f1{tag}"] = cm, m["iou"], np.array(m["miou"]), m["f1"]
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        loss, _, _ = R.train_step(sd, opt, xA[s:s + BATCH], xB[s:s + BATCH], mask[s:s + BATCH])
        losses.append(loss)
        print(f"step {k}: loss {loss:.6f}", flush=True)
        if k + 1 in CHECKPOINTS:
            evaluate(str(k + 1))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "snunet_parity_run.npz"), losses=np.array(losses),
                        protocol=np.array([K_STEPS, TRAIN_TILES, BATCH, HELD_OUT, HELD_OUT_SEED, TRAIN_SEED]), **out)


def main_reference():
    """the protocol of main() driven through the REAL reference modules (build container only): model(xA, xB) -> BCEandDiceLoss ->
    backward -> Adam.step as training/change_detection_trainer.py:135-180 does; eval-mode inference + the integer metrics of
    oracle/metrics_ref.py (torchmetrics is not installed) on the held-out tiles"""
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from models.snunet import SNUNet_ECAM                      # (reference)
    from utilities.bce_and_dice import BCEandDiceLoss          # (reference)
    torch.set_num_threads(8)
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()
    model = SNUNet_ECAM(2, 3, base_channel=32)
    seeded_fill_(model.state_dict())
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    criterion = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    losses, out = [], {}

    def evaluate(tag):
        cm = np.zeros((4, 4), np.int64)
        model.eval()
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                logits = model(eA[s:s + 8], eB[s:s + 8])
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits.numpy()), emask[s:s + 8].numpy())
        model.train()
        m = metrics_ref.metrics_from_cm(cm)
        print(tag, "cm\n", cm, "\niou", m["iou"], "miou", m["miou"], flush=True)
        out[f"cm{tag}"], out[f"iou{tag}"], out[f"miou{tag}"], out[f"f1{tag}"] = cm, m["iou"], np.array(m["miou"]), m["f1"]
    model.train()
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        optimizer.zero_grad()
        loss = criterion(model(xA[s:s + BATCH], xB[s:s + BATCH]), mask[s:s + BATCH])
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        print(f"step {k}: loss {losses[-1]:.6f}", flush=True)
        if k + 1 in CHECKPOINTS:
            evaluate(str(k + 1))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "snunet_parity_run_ref.npz"), losses=np.array(losses),
                        protocol=np.array([K_STEPS, TRAIN_TILES, BATCH, HELD_OUT, HELD_OUT_SEED, TRAIN_SEED]), **out)


def main_reference_plateau():
    """the PLATEAU protocol (see the constants) on the imported reference -> tests/golden/snunet_parity_plateau_ref.npz
    (build container only; ~25 minutes on 8 CPU threads, ~30 GB)"""
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from models.snunet import SNUNet_ECAM                      # (reference)
    from utilities.bce_and_dice import BCEandDiceLoss          # (reference)
    torch.set_num_threads(8)
    (xA, xB), mask = cd_inputs(make_batch(P_TRAIN_TILES, seed=TRAIN_SEED), ("pre_event_1", "post_event"))
    (eA, eB), emask = cd_inputs(make_batch(HELD_OUT, seed=HELD_OUT_SEED), ("pre_event_1", "post_event"))
    model = SNUNet_ECAM(2, 3, base_channel=32)
    seeded_fill_(model.state_dict())
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    criterion = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    losses, out = [], {}

    def evaluate(tag):
        cm = np.zeros((4, 4), np.int64)
        model.eval()
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                logits = model(eA[s:s + 8], eB[s:s + 8])
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(logits.numpy()), emask[s:s + 8].numpy())
        model.train()
        m = metrics_ref.metrics_from_cm(cm)
        print(tag, "cm\n", cm, "\niou", m["iou"], "miou", m["miou"], flush=True)
        out[f"cm{tag}"], out[f"iou{tag}"], out[f"miou{tag}"], out[f"f1{tag}"] = cm, m["iou"], np.array(m["miou"]), m["f1"]
    model.train()
    for k in range(P_K_STEPS):
        s = (k % (P_TRAIN_TILES // P_BATCH)) * P_BATCH
        optimizer.zero_grad()
        loss = criterion(model(xA[s:s + P_BATCH], xB[s:s + P_BATCH]), mask[s:s + P_BATCH])
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        print(f"step {k}: loss {losses[-1]:.6f}", flush=True)
        if k + 1 in P_CHECKPOINTS or (k + 1) % 10 == 0:
            evaluate(str(k + 1))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "snunet_parity_plateau_ref.npz"), losses=np.array(losses),
                        protocol=np.array([P_K_STEPS, P_TRAIN_TILES, P_BATCH, HELD_OUT, HELD_OUT_SEED, TRAIN_SEED]), **out)


def main_changeformer():
    """The same protocol for ChangeFormerV6 (2-band tiles) on the imported reference: the shipped optimiser of the method
    (configs/method/changeformer/changeformer.json: SGD lr 6e-4, momentum 0.99, weight decay 1e-5; change_detection_trainer.py:45-66),
    ce+dice on the sigmoid map (decoder_softmax), the stochastic layers at p = 0 on both sides (their draws come from different
    generators: tests/golden/changeformer_bench.npz pins them on the counter-based stream instead).
    -> tests/golden/changeformer_parity_run_ref.npz"""
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    from oracle.gen_golden import _import_changeformer_reference
    from utilities.bce_and_dice import BCEandDiceLoss          # (reference)
    ChangeFormerV6 = _import_changeformer_reference()
    torch.set_num_threads(8)
    (xA, xB, mask), (eA, eB, emask) = protocol_tiles()
    model = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=256)
    seeded_fill_(model.state_dict())
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    optimizer = torch.optim.SGD(model.parameters(), lr=6e-4, momentum=0.99, weight_decay=1e-5)
    criterion = BCEandDiceLoss(weights=[1.0, 1.0, 1.0], ignore_index=3, use_softmax=True)
    losses, out = [], {}

    def evaluate(tag):
        cm = np.zeros((4, 4), np.int64)
        model.eval()
        with torch.no_grad():
            for s in range(0, HELD_OUT, 8):
                prob = model(eA[s:s + 8], eB[s:s + 8])[-1]
                cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(prob.numpy()), emask[s:s + 8].numpy())
        model.train()
        m = metrics_ref.metrics_from_cm(cm)
        print(tag, "cm\n", cm, "\niou", m["iou"], "miou", m["miou"], flush=True)
        out[f"cm{tag}"], out[f"iou{tag}"], out[f"miou{tag}"], out[f"f1{tag}"] = cm, m["iou"], np.array(m["miou"]), m["f1"]
    model.train()
    for k in range(K_STEPS):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        optimizer.zero_grad()
        loss = criterion(model(xA[s:s + BATCH], xB[s:s + BATCH])[-1], mask[s:s + BATCH])
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        print(f"step {k}: loss {losses[-1]:.6f}", flush=True)
        if k + 1 in CHECKPOINTS:
            evaluate(str(k + 1))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "changeformer_parity_run_ref.npz"), losses=np.array(losses),
                        protocol=np.array([K_STEPS, TRAIN_TILES, BATCH, HELD_OUT, HELD_OUT_SEED, TRAIN_SEED]), **out)


if __name__ == "__main__":
    if "--changeformer" in sys.argv[1:]:
        main_changeformer()
    elif "--reference-plateau" in sys.argv[1:]:
        main_reference_plateau()
    elif "--reference" in sys.argv[1:]:
        main_reference()
    else:
        main()
