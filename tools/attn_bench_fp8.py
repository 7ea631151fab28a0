import torch
from loongx_amd import ops
import sys
dev = "cuda"; B, H = 1, 24; lens = (512, 4096, 4096) if len(sys.argv) > 1 else (512, 1024, 1024); D = H * 128
M = B * sum(lens)
buf = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
row0 = [0, B * lens[0], B * (lens[0] + lens[1])]; vt0 = [0, lens[0], lens[0] + lens[1]]
Q8 = torch.zeros(M, D, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8)
VT8 = torch.zeros(B, H, 128, sum(lens), dtype=torch.uint8, device=dev)
segs = [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)]
O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
def prep(): ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
def run(): ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0)
prep()
for name, fn in (("qkv_prep_fp8", prep), ("attn_fp8", run)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 50
    S = sum(lens)
    print(f"{name} {us:.1f} us" + (f"  {4*B*H*S*S*128/us/1e6:.0f} TF" if name == "attn_fp8" else ""))
