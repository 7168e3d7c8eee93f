import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kurosiwo_amd import _lib
from kurosiwo_amd.runtime import stream_ptr
lib=_lib.load(); dev=torch.device("cuda:0"); torch.manual_seed(2)
def r(a,ref): return float((a.double()-ref).norm()/ref.norm())
for rows,K,N in [(3136,128,512),(196,512,1024),(784,320,1280),(3136,512,128),(12544,128,128)]:
    # strided x (view into a wider matrix), strided out, residual with its own stride
    X=(torch.randn(rows,K+64,device=dev)).bfloat16(); x=X[:,32:32+K]
    w=(torch.randn(N,K,device=dev)*K**-0.5).bfloat16(); b=torch.randn(N,device=dev)
    R=(torch.randn(rows,N+16,device=dev)).bfloat16(); res=R[:,8:8+N]
    Y=torch.zeros(rows,N+32,device=dev,dtype=torch.bfloat16); y=Y[:,16:16+N]
    _lib.check(lib.ksmi_gemm_nt(x.data_ptr(),K+64,w.data_ptr(),K,b.data_ptr(),res.data_ptr(),N+16,y.data_ptr(),N+32,rows,K,N,stream_ptr()),"nt")
    ref=x.double()@w.double().t()+b.double()+res.double()
    e1=r(y,ref); pad=float(Y[:,:16].abs().max()+Y[:,16+N:].abs().max())
    # nn with accumulate, strided dy
    DY=(torch.randn(rows,N+64,device=dev)).bfloat16(); dy=DY[:,64:]
    base=(torch.randn(rows,K,device=dev)).bfloat16(); dx=base.clone()
    _lib.check(lib.ksmi_gemm_nn(dy.data_ptr(),N+64,w.data_ptr(),K,dx.data_ptr(),K,rows,K,N,1,stream_ptr()),"nn")
    refd=dy.double()@w.double()+base.double()
    print(rows,K,N,"nt+resid strided",f"{e1:.2e}","pad",pad,"nn acc",f"{r(dx,refd):.2e}")
