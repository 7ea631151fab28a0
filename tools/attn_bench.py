import torch, math
from loongx_amd import ops
import sys
dev="cuda"; B,H=1,24; lens=(512,4096,4096) if len(sys.argv)>1 else (512,1024,1024); D=H*128
M=B*sum(lens)
buf=torch.randn(M,3*D,device=dev).to(torch.bfloat16)
row0=[0,B*lens[0],B*(lens[0]+lens[1])]; vt0=[0,lens[0],lens[0]+lens[1]]
VT=torch.zeros(B,H,128,sum(lens),dtype=torch.bfloat16,device=dev)
ops.qkv_prep_segs(buf,2*D,0,D,[(row0[i],lens[i],vt0[i],None,None,None,None) for i in range(3)],B,H,VT)
obuf=torch.zeros(M,D,dtype=torch.bfloat16,device=dev)
def run(): ops.attn_fwd(buf,buf,VT,obuf,q_col=2*D,k_col=0,o_col=0,B=B,H=H,seg_row0=row0,seg_len=list(lens),seg_vt0=vt0)
for _ in range(5): run()
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
us=s.elapsed_time(e)*1e3/50
S=sum(lens); print(f"attn {us:.1f} us  {4*B*H*S*S*128/us/1e6:.0f} TF")
