"""GEMM microbench: TFLOP/s for the DiT's GEMM shapes (B=1, S=2560). Usage: gemm_bench.py [iters]"""
import sys, os, time
import torch
from loongx_amd import ops
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
D = 3072
shapes = [("qkv+mlp fused", 2560, 7 * D, D, "bf16"), ("ff1", 2560, 4 * D, D, "gelu"), ("qkv", 2560, 3 * D, D, "bf16"),
          ("out", 2560, D, D, "resid"), ("ff2", 2560, D, 4 * D, "resid"), ("single out", 2560, D, 5 * D, "resid")]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    if epi == "resid":
        C = torch.zeros(M, N, device=dev); gate = torch.ones(1, N, device=dev)
        d = ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate)
    elif epi == "gelu":
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        d = ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)
    else:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        d = ops.gemm_desc(A, W, C, bias=bias)
    for bm in ([256, 128] if os.environ.get("BOTH") else [0]):
        if bm: os.environ["LX_GEMM_BM"] = str(bm)
        for _ in range(3): ops.gemm([d])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): ops.gemm([d])
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / it
        print(f"{name:14s} M={M} N={N:6d} K={K:6d} bm={bm or 'auto':>4} {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
