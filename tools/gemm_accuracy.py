import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sys
from kurosiwo_amd import functional as Fk
dev=torch.device("cuda:0"); torch.manual_seed(1)
shapes=[(12544,128,128),(3136,128,512),(3136,512,128),(784,320,1280),(784,1280,320),(196,512,512),(196,512,2048),(196,2048,512),(196,512,1024),(6272,128,256),(1568,320,256),(392,512,256),(3136,128,256)]
for rows,K,N in shapes:
    x=(torch.randn(rows,K,device=dev)).bfloat16(); w=(torch.randn(N,K,device=dev)*K**-0.5).bfloat16(); b=torch.randn(N,device=dev)
    dy=(torch.randn(rows,N,device=dev)).bfloat16()
    r=lambda a,ref: float((a.double()-ref).norm()/ref.norm())
    m=lambda a,ref: float((a.double()-ref).abs().max()/ref.abs().max())
    ref=x.double()@w.double().t()+b.double()
    y=Fk.gemm_nt(x,w,b)
    refd=dy.double()@w.double(); dx=Fk.gemm_nn(dy,w)
    refw=dy.double().t()@x.double(); dw=Fk.linear_wgrad(x,dy)
    print(f"{rows:6d} {K:5d} {N:5d}  nt l2 {r(y,ref):.2e} max {m(y,ref):.2e} | nn l2 {r(dx,refd):.2e} max {m(dx,refd):.2e} | wgrad l2 {r(dw,refw):.2e} max {m(dw,refw):.2e}")
