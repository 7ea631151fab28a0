#!/usr/bin/env python3
"""The GEMM priced in joules (round-5 review, item 5): the part runs the denoise step at its 1400 W package cap, so a change that saves
cycles but not energy saves no time. Back-to-back launches of the LARGEST launch of the step -- the single-block fused projection,
[k | v | q | mlp] = 21504 x 3072, with its real epilogue (LX_EPI_QKV-free form: 16-bit store + GELU columns) -- at batch 1 (2560 rows) and
batch 16 (40960 rows), >= 2000 launches per arm (batch 16: >= 300), with the amdgpu hwmon power / clock sampled every 50 ms in-process:
us per launch, sustained clock, average package power, JOULES per launch and pJ per algorithmic FLOP.
Arms are libraries / environment: the shipped plan, the 8-wave 32x32x16 kernels on the same 256x256x64 tile (LX_GEMM4=0), and tile-patch
heights GROUP_M = 2 / 8 / 16 (variant libraries: an XCD's 32 concurrent tiles as 2 x 16, 4 x 8 (shipped), 8 x 4, 16 x 2 panels).
    python tools/gemm_energy.py <arm name>          (one arm per process: the library and its environment are read once)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from loongx_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "shipped"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, K = 21504, 3072
W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
bias = torch.randn(N, device=dev, generator=g) * 0.02
ws = ops.gemm_workspace(dev)
for M, launches in ((2560, 2400), (40960, 320)):
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)          # random operands: what the matrix pipe toggles on
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, gelu_col_start=9216, rows_per_batch=M)
    for _ in range(20):
        ops.gemm([d], ws)
    torch.cuda.synchronize()
    time.sleep(0.3)
    p = bench.PowerSampler(0)
    p.start()
    t0 = time.perf_counter()
    for _ in range(launches):
        ops.gemm([d], ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = p.stop() or {}
    us = dt / launches * 1e6
    fl = 2.0 * M * N * K
    W_ = rec.get("avg_W")
    print(f"{name:10s} M={M:6d}: {us:8.1f} us/launch {fl / us / 1e6:7.1f} TFLOP/s  sclk {rec.get('sclk_MHz_avg')} MHz  {W_} W  "
          f"{(W_ * us * 1e-6) if W_ else float('nan'):7.4f} J/launch  {(W_ * us * 1e-6 / fl * 1e12) if W_ else float('nan'):6.3f} pJ/FLOP  ({rec.get('samples')} samples)", flush=True)
    del A, C
