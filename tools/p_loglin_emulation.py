"""CPU emulation (torch, e4m3 via float8_e4m3fn) of the probability encodings of lx_attn_fp8_pipe_kernel, tile by tile with the kernel's deferred-rescale
reference rule: shipped round 5 (true exp2 + e4m3 rounding, reference at 2^0, threshold 8), round 6 with the 2^6 factor (threshold 2), and the
log-linear byte code. Errors are relative L2 of the attention output against fp32 softmax attention on the unrounded q / k / v.
    python tools/p_loglin_emulation.py   (no GPU; ~2 min)"""
import torch, math
torch.manual_seed(0)
def e4m3(x): return x.to(torch.float8_e4m3fn).float()
def decode_byte(b):
    b=b.to(torch.int32); e=(b>>3)&15; m=b&7
    return torch.where(e==0, m.float()*2.0**-9, (1+m.float()/8)*torch.exp2(e.float()-7))
def attn_tiled(sc, v8, mode, thr, offs):
    # sc: [H,Sq,S] log2-unit scores (from e4m3 q,k). emulate per-row deferred reference: ref starts at tile-0 max, moves up by excess when tile max > ref+thr
    H,Sq,S=sc.shape
    ref=sc[:,:,:64].max(-1,keepdim=True).values
    O=torch.zeros(H,Sq,128); L=torch.zeros(H,Sq,1)
    for t0 in range(0,S,64):
        s=sc[:,:,t0:t0+64]
        tm=s.max(-1,keepdim=True).values
        # wave-level decision in the kernel (any row of 32 exceeds) -> approximate per row
        exc=(tm-ref).clamp_min(0)
        move=(tm-ref)>thr
        d=torch.where(move,exc,torch.zeros_like(exc))
        a=torch.exp2(-d); O*=a; L*=a; ref=ref+d
        x=s-ref+offs
        if mode=='true': P=e4m3(torch.exp2(x))
        else: P=decode_byte((8*x+56.5).clamp(0,255).floor())
        O+=P@v8[:,t0:t0+64]; L+=P.sum(-1,keepdim=True)
    return O/L
def run(S,gain,H=2,Sq=128,peaky=False):
    q=torch.randn(H,Sq,128); k=torch.randn(H,S,128); v=torch.randn(H,S,128)
    if peaky:
        k[:,::97]*=2.0
    scf=(q@k.transpose(1,2))/math.sqrt(128)*gain
    ref=torch.softmax(scf,-1)@v
    q8=e4m3(q*16.32/ (1.0))/16.32; k8=e4m3(k*16)/16; v8=e4m3(v)
    sc=(q8@k8.transpose(1,2))/math.sqrt(128)*gain*1.4426950408889634
    outs={'shipped(true,thr8,off0)':attn_tiled(sc,v8,'true',8,0),'true,thr2,off6':attn_tiled(sc,v8,'true',2,6),'loglin,thr2,off6':attn_tiled(sc,v8,'lin',2,6),
          'true,thr4,off4':attn_tiled(sc,v8,'true',4,4)}
    P=torch.softmax(sc*math.log(2),-1); outs['exactP']=P@v8
    return {n:round(float(((o-ref).norm()/ref.norm())),5) for n,o in outs.items()}
for S in (2560,8704):
    for gain in (1.0,2.0,3.0):
        for pk in (False,True):
            print(S,gain,pk,run(S,gain,peaky=pk))
