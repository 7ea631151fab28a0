"""qkv_prep timing at the FLUX.1-dev shape with norm weights and RoPE tables (LX_PREP_HPB=1|2|4 selects heads per block)."""
import torch
from loongx_amd import ops
dev = "cuda"; B, H = 1, 24; lens = (512, 1024, 1024); D = H * 128
M = B * sum(lens)
buf = torch.randn(M, 7 * D, device=dev).to(torch.bfloat16)
row0 = [0, B * 512, B * 1536]; vt0 = [0, 512, 1536]
VT = torch.zeros(B, H, 128, 2560, dtype=torch.bfloat16, device=dev)
w = torch.ones(128, device=dev)
tabs = [(torch.rand(L, 128, device=dev), torch.rand(L, 128, device=dev)) for L in lens]
segs = [(row0[i], lens[i], vt0[i], w, w, tabs[i][0], tabs[i][1]) for i in range(3)]
def run(): ops.qkv_prep_segs(buf, 2 * D, 0, D, segs, B, H, VT)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): run()
e.record(); torch.cuda.synchronize()
print(f"qkv_prep {s.elapsed_time(e) * 1e3 / 100:.1f} us")
