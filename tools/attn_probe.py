#!/usr/bin/env python3
"""Where a wave of lx_attn_pipe_kernel spends its cycles (measurement build: tools/build_variant.sh probe attn -DLX_ATTN_PROBE=1,
run with LX_AMD_LIB=loongx_amd/lib/liblx_amd_probe.so): kernel entry -> exit, the tile loop, and inside it the end-of-iteration
`s_waitcnt vmcnt(0)` (own LDS-DMA pieces) and `s_barrier` (the other seven waves)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from loongx_amd import ops
from loongx_amd._lib import lib

big = "--big" in sys.argv
fp8 = "--fp8" in sys.argv
dev = "cuda"; B, H = 1, 24; lens = (512, 4096, 4096) if big else (512, 1024, 1024); D = H * 128
M = B * sum(lens)
g = torch.Generator(device=dev).manual_seed(0)
buf = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
row0 = [0, B * lens[0], B * (lens[0] + lens[1])]; vt0 = [0, lens[0], lens[0] + lens[1]]
segs = [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)]
O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
VT = torch.zeros(B, H, 128, sum(lens), dtype=torch.bfloat16, device=dev)
ops.qkv_prep_segs(buf, 2 * D, 0, D, segs, B, H, VT)
run = lambda: ops.attn_fwd(buf, buf, VT, O, q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0)
if fp8:
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8)
    VT8 = torch.zeros(B, H, 128, sum(lens), dtype=torch.uint8, device=dev)
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
    run = lambda: ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0)
for _ in range(5):
    run()
torch.cuda.synchronize()
n_wg = H * sum((L + 255) // 256 for L in lens)
n = min(n_wg, 4096) * 8 * 4
host = (ctypes.c_ulonglong * n)()
lib.lx_attn_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_size_t]
assert lib.lx_attn_probe_read(host, n) == 0
a = np.array(host, dtype=np.float64).reshape(-1, 8, 4)
tiles = sum(lens) / 64
tot, loop, vm, bar = (a[..., i] for i in range(4))
print(f"workgroups {a.shape[0]}, key tiles per workgroup {tiles:.0f}  (s_memtime ticks; 100 MHz constant clock => x (shader MHz / 100) for shader cycles)")
for name, w in (("waves 0-3 (older)", slice(0, 4)), ("waves 4-7 (younger)", slice(4, 8))):
    print(f"  {name}: kernel {tot[:, w].mean():9.0f}  loop {loop[:, w].mean():9.0f} ({loop[:, w].mean() / tot[:, w].mean():.3f})  "
          f"per tile {loop[:, w].mean() / tiles:7.1f}  vmcnt wait {vm[:, w].mean():8.0f} ({vm[:, w].mean() / loop[:, w].mean():.3f} of loop)  "
          f"barrier {bar[:, w].mean():8.0f} ({bar[:, w].mean() / loop[:, w].mean():.3f} of loop)")
print(f"  prologue + epilogue share of the kernel: {1 - loop.mean() / tot.mean():.3f}")
