import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from kurosiwo_amd.fcsiam import SiamUnet_conc, SiamUnet_diff
from kurosiwo_amd.loss import BCEandDiceLoss
for cls in (SiamUnet_conc, SiamUnet_diff):
    res={}
    for prec in ("fp32","bf16"):
        torch.manual_seed(0)
        m=cls(2,3,precision=prec).cuda().train(); m.manual_seed(5,0)
        g=torch.Generator().manual_seed(1)
        x1=torch.randn(8,2,224,224,generator=g).cuda(); x2=torch.randn(8,2,224,224,generator=g).cuda(); y=torch.randint(0,3,(8,224,224),generator=g).cuda()
        out=m(x1,x2); loss=BCEandDiceLoss(weights=[1,1,1],ignore_index=3,use_softmax=True)(out,y); loss.backward()
        res[prec]=(float(loss), {k:p.grad.float().cpu().double() for k,p in m.named_parameters()}, out.detach().float().cpu())
    cos=[]
    for k in res["fp32"][1]:
        a,b=res["fp32"][1][k],res["bf16"][1][k]
        if float(a.norm())==0: continue
        cos.append(float((a*b).sum()/(a.norm()*b.norm()+1e-30)))
    cos=np.array(cos)
    print(cls.__name__, "loss", res["fp32"][0], res["bf16"][0], "out maxdiff", float((res["fp32"][2]-res["bf16"][2]).abs().max()), "cos median %.4f min %.4f frac>0.99 %.2f"%(np.median(cos), cos.min(), (cos>0.99).mean()))
