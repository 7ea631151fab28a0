"""Interleaved A/B of GEMM variants inside ONE process (variants = (env LX_GEMM_BM, LX_GEMM_V, epilogue debug OR-mask))."""
import os, sys
import torch
from loongx_amd import ops
dev = "cuda"
D = 3072
shapes = [("fused", 2560, 7 * D, D), ("ff1", 2560, 4 * D, D), ("qkv", 2560, 3 * D, D), ("out", 2560, D, D), ("ff2", 2560, D, 4 * D), ("sout", 2560, D, 5 * D)]
variants = [v.split(":") for v in (sys.argv[1] if len(sys.argv) > 1 else "0:1:0,0:1:0x2000").split(",")]   # bm:ver:epi_or
rounds = int(os.environ.get("ROUNDS", 5)); it = int(os.environ.get("IT", 10))
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {i: [] for i in range(len(variants))}
    for r in range(rounds):
        for i, (bm, ver, eo) in enumerate(variants):
            os.environ["LX_GEMM_BM"] = bm; os.environ["LX_GEMM_V"] = ver
            d = ops.gemm_desc(A, W, C); d.epilogue |= int(eo, 0)
            ops.gemm([d]); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(it): ops.gemm([d])
            e.record(); torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) * 1e3 / it)
    line = f"{name:6s} N={N:6d} K={K:6d} "
    for i, v in enumerate(variants):
        us = sorted(res[i])[len(res[i]) // 2]
        line += f"| {':'.join(v):>12s} {us:7.1f}us {2*M*N*K/us/1e6:6.0f}TF "
    print(line)
