#!/usr/bin/env python3
"""Run the bf16 attention launch N times on one shape (a driver for rocprofv3 passes):
    python tools/attn_run.py [--big | --shape BxHxL] [--iters N] [--flags F]     (flags 3 = the bounded-score contract, the engine's default)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongx_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--big", action="store_true")
ap.add_argument("--shape", default="")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--flags", type=int, default=3)
ap.add_argument("--fp8", action="store_true", help="lx_attn_fwd_fp8 (flags then: 0 = log-linear probability bytes, 32 = LX_ATTN_P_EXP2)")
a = ap.parse_args()
if a.fp8 and a.flags == 3:
    a.flags = 0
dev = "cuda"
B, H = 1, 24
lens = (512, 4096, 4096) if a.big else (512, 1024, 1024)
if a.shape:
    B, H, L0 = (int(v) for v in a.shape.split("x"))
    lens = (L0,)
D = H * 128
M = B * sum(lens)
g = torch.Generator(device=dev).manual_seed(0)
buf = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
row0 = [B * sum(lens[:i]) for i in range(len(lens))]
vt0 = [sum(lens[:i]) for i in range(len(lens))]
one = torch.ones(128, device=dev)
oneq = one * ops.Q_LOG2_FACTOR if a.flags else one
segs = [(row0[i], lens[i], vt0[i], oneq, one, None, None) for i in range(len(lens))]
O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
VT = torch.zeros(B, H, 128, sum(lens), dtype=torch.bfloat16, device=dev)
q = buf.clone()
ops.qkv_prep_segs(q, 2 * D, 0, D, segs, B, H, VT)
run = lambda: ops.attn_fwd(q, q, VT, O, q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=a.flags)
if a.fp8:
    segs8 = [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(len(lens))]
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8)
    VT8 = torch.zeros(B, H, 128, sum(lens), dtype=torch.uint8, device=dev)
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs8, B, H, Q8, K8, VT8)
    run = lambda: ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=a.flags)
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / a.iters
S = sum(lens)
print(f"attention B={B} H={H} S={S}: {us:.1f} us per launch, {4 * B * H * S * S * 128 / us / 1e6:.0f} TFLOP/s")
