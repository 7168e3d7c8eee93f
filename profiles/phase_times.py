#!/usr/bin/env python3
"""Per-workgroup phase timestamps of the igemm2 kernel (KSMI_DBG=8): prologue / K loop / epilogue cycles."""
import os, sys
os.environ["KSMI_DBG"] = "8"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kurosiwo_amd import functional as Fk
dev = torch.device("cuda:0")
for name, H, cs, N in [("L0 conv0_4", 224, [32] * 5 + [64], 32), ("L0 K=32", 224, [32], 32), ("L1 conv1_1", 112, [64, 64, 128], 64)]:
    xs = [(torch.randn(32, H, H, c, device=dev) * 0.5).to(torch.bfloat16) for c in cs]
    w = torch.randn(N, sum(cs), 3, 3, device=dev) * 0.05
    for _ in range(2):
        y, st = Fk.conv3x3(xs, w, None, want_stats=True)
    torch.cuda.synchronize()
    t = st.view(torch.int64).reshape(-1)[: st.shape[0] * 8].reshape(-1, 8).cpu()
    t = t[t[:, 0] > 0]
    pro, loop, epi = (t[:, 1] - t[:, 0]).float(), (t[:, 2] - t[:, 1]).float(), (t[:, 3] - t[:, 2]).float()
    issue = (t[:, 4] - t[:, 2]).float()
    span = float(t[:, 3].max() - t[:, 0].min())
    print(f"{name}: blocks {len(t)} prologue {pro.median():.0f} loop {loop.median():.0f} epilogue {epi.median():.0f} (issue {issue.median():.0f}) cycles (median); "
          f"kernel span {span:.0f} ticks; sum/block {float((t[:,3]-t[:,0]).float().median()):.0f}")
