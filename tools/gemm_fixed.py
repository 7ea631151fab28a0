"""What is the fixed (K-independent) cost of the wide GEMM made of? Small-K launches with bf16 vs fp32 stores, and N halved."""
import torch
from loongx_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timed(d, it=30):
    for _ in range(3): ops.gemm([d])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): ops.gemm([d])
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
for M, N in ((2560, 21504), (2560, 10752), (4096, 4096)):
    for K in (64, 256, 1024):
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
        Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cf = torch.empty(M, N, device=dev, dtype=torch.float32)
        tb = min(timed(ops.gemm_desc(A, W, Cb)) for _ in range(3))
        tf = min(timed(ops.gemm_desc(A, W, Cf, epilogue=ops.LX_EPI_STORE_F32)) for _ in range(3))
        print(f"M={M} N={N} K={K:5d}: bf16 store {tb:6.1f} us ({M*N*2/tb/1e6:5.2f} TB/s of output) | fp32 store {tf:6.1f} us ({M*N*4/tf/1e6:5.2f} TB/s)")
