"""Hot vs cold-weight GEMM timing: cycle through NW distinct weight buffers (NW*bytes >> 256 MB Infinity Cache)."""
import os, sys
import torch
from loongx_amd import ops
dev = "cuda"
D = 3072
M = 2560
shapes = [("out", D, D), ("ff2", D, 4 * D), ("sout", D, 5 * D), ("ff1", 4 * D, D), ("fused", 7 * D, D)]
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in shapes:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    nw = max(2, int(1.2e9 // (N * K * 2)))
    Ws = [ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    bias = torch.zeros(N, device=dev)
    resid = N == D
    if resid:
        C = torch.zeros(M, N, device=dev); gate = torch.ones(1, N, device=dev)
        mk = lambda W: ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate)
    else:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        mk = lambda W: ops.gemm_desc(A, W, C, bias=bias)
    for mode in ("hot", "cold"):
        ds = [mk(Ws[0])] * nw if mode == "hot" else [mk(W) for W in Ws]
        for d in ds: ops.gemm([d])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        s.record()
        for _ in range(reps):
            for d in ds: ops.gemm([d])
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / (reps * nw)
        print(f"{name:6s} N={N:6d} K={K:6d} {mode:4s} nw={nw:3d} {us:8.1f} us {2*M*N*K/us/1e6:7.0f} TF   W={N*K*2/1e6:.0f} MB")
