import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kurosiwo_amd.changeformer import ChangeFormerV6
from kurosiwo_amd.trainer import CDTrainStep
torch.manual_seed(1)
m = ChangeFormerV6(input_nc=2, output_nc=3, decoder_softmax=True, embed_dim=256, precision="bf16").cuda().train()
for B in (1, 3, 5, 6, 7):
    st = CDTrainStep(m, B, 224, 224, "ce+dice", (1.0, 1.0, 1.0), lr=1e-4)
    g = torch.Generator().manual_seed(B)
    xa, xb = torch.randn(B, 2, 224, 224, generator=g), torch.randn(B, 2, 224, 224, generator=g)
    y = torch.randint(0, 3, (B, 224, 224), generator=g)
    l = st.step(xa.cuda(), xb.cuda(), y.cuda()); torch.cuda.synchronize()
    m.eval()
    with torch.no_grad():
        out = m(xa.cuda(), xb.cuda())
    m.train()
    print("B", B, "loss", [round(float(v), 4) for v in l], "eval out", tuple(out[-1].shape), bool(torch.isfinite(out[-1]).all()))
