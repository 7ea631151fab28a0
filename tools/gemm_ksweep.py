"""K sweep at fixed output size: separates the per-tile fixed cost from the per-K-tile cost (read with rocprofv3 kernel trace)."""
import os, sys
import torch
from loongx_amd import ops
dev = "cuda"
M = int(os.environ.get("M", 2560)); N = int(os.environ.get("N", 3072))
g = torch.Generator(device=dev).manual_seed(0)
for K in (64, 512, 1024, 2048, 3072, 6144, 12288):
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = ops.gemm_desc(A, W, C)
    d.epilogue |= int(os.environ.get('EPI_OR', '0'), 0)
    for _ in range(6):
        ops.gemm([d])
    torch.cuda.synchronize()
