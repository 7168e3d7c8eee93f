import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.trainer import CDTrainStep
from oracle import metrics_ref, snunet_ref as R
from oracle.gen_parity_run import BATCH, HELD_OUT, K_STEPS, TRAIN_TILES, protocol_tiles
from oracle.seeded import seeded_fill_
dev = torch.device("cuda:0")
(xA, xB, mask), (eA, eB, emask) = protocol_tiles()
K = int(os.environ.get("K", K_STEPS))
def evaluate(sd, prec):
    m = SNUNet_ECAM(2, 3, base_channel=32, precision=prec); m.load_state_dict(sd); m = m.to(dev).eval()
    cm = np.zeros((4, 4), np.int64)
    with torch.no_grad():
        for s in range(0, HELD_OUT, 8):
            lg = m(eA[s:s+8].to(dev), eB[s:s+8].to(dev)).float().cpu().numpy()
            cm += metrics_ref.confusion_matrix(metrics_ref.argmax_lowest_index(lg), emask[s:s+8].numpy())
    return metrics_ref.metrics_from_cm(cm)
for tp in ("fp32", "bf16"):
    model = SNUNet_ECAM(2, 3, base_channel=32, precision=tp); model.load_state_dict(seeded_fill_(R.new_state_dict(2, 3, 32))); model = model.to(dev).train()
    step = CDTrainStep(model, BATCH, 224, 224, loss_function="ce+dice", class_weights=(1.0, 1.0, 1.0), lr=1e-3)
    ls = []
    for k in range(K):
        s = (k % (TRAIN_TILES // BATCH)) * BATCH
        ls.append(float(step.step(xA[s:s+BATCH].to(dev), xB[s:s+BATCH].to(dev), mask[s:s+BATCH].to(dev))[0]))
        if (k + 1) in (10, 20, 40, 80, 160):
            sd = {kk: v.detach().cpu().clone() for kk, v in model.state_dict().items()}
            for ep in ("fp32", "bf16"):
                mm = evaluate(sd, ep)
                print(f"train {tp} K={k+1} eval {ep}: miou {mm['miou']:.5f} iou {np.array2string(mm['iou'][:3], precision=4)} loss {ls[-1]:.5f}", flush=True)
