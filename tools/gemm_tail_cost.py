"""What does the partial last round of a launch cost? The fused single-block projection's shape (M = 2560, K = 3072, bf16 store) at N = 256 n
for n around the round boundaries (10 n tiles on 256 CUs): time per launch (with the caller's workspace: the planner's split tail is live)
against the tile count -- back-to-back launches, alternating sweeps, minimum of the medians.
    python tools/gemm_tail_cost.py [--iters 30]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongx_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--M", type=int, default=2560)
ap.add_argument("--K", type=int, default=3072)
a = ap.parse_args()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, K = a.M, a.K
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
ws = ops.gemm_workspace(dev)
NS = [51, 52, 64, 76, 77, 78, 80, 84, 90, 96, 102, 103]          # column tiles: 10 n tiles each
probs = {}
for n in NS:
    N = 256 * n
    W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    probs[n] = ops.gemm_desc(A, W, C, bias=torch.zeros(N, device=dev), epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)
res = {n: [] for n in NS}
for rep in range(4):
    for n in (NS if rep % 2 == 0 else NS[::-1]):
        for _ in range(3):
            ops.gemm([probs[n]], ws)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            ops.gemm([probs[n]], ws)
        e.record()
        torch.cuda.synchronize()
        res[n].append(s.elapsed_time(e) * 1e3 / a.iters)
ncu = 256
for n in NS:
    t = min(res[n])
    tiles = (M // 256) * n
    print(f"N = 256 x {n:3d}: {tiles:4d} tiles = {tiles / ncu:5.2f} rounds ({tiles % ncu:3d} in the last): {t:7.1f} us = {t / (tiles / ncu):6.1f} us per round-equivalent, "
          f"{2 * M * 256 * n * K / t / 1e6:7.1f} TFLOP/s")
