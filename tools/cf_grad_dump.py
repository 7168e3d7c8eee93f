import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from kurosiwo_amd.changeformer import ChangeFormerV6
from kurosiwo_amd.loss import BCEandDiceLoss
from oracle import changeformer_ref as R
from oracle.seeded import seeded_fill_, seeded_tensor, seeded_labels
m = ChangeFormerV6(2, 3, decoder_softmax=True, embed_dim=256, precision="bf16")
m.drop_rate = m.attn_drop = m.drop_path_rate = 0.0
m.load_state_dict(seeded_fill_(R.new_state_dict(2, 3, 256))); m = m.cuda().train()
x1 = seeded_tensor("changeformer.train.x1", (2, 2, 224, 224)).clamp_(-2.23, 5.75).cuda()
x2 = seeded_tensor("changeformer.train.x2", (2, 2, 224, 224)).clamp_(-2.23, 5.75).cuda()
lbl = seeded_labels("changeformer.train.lbl", (2, 224, 224)).cuda()
outs = m(x1, x2)
loss = BCEandDiceLoss(weights=[0.3715753140309927, 14.009780283125977, 8.20405370357821], ignore_index=3, use_softmax=True)(outs[-1], lbl)
loss.backward()
d = {k: p.grad.float().cpu().numpy() for k, p in m.named_parameters()}
d["__out"] = outs[-1].detach().float().cpu().numpy()
np.savez(sys.argv[1], **d)
