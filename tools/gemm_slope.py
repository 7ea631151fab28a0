"""Per-K-tile cost and fixed cost of exactly one full round of tiles (256 workgroups), 256- and 128-row tiles: K sweep timed with
HIP events, back to back, minimum of 3 passes. Pre-tiled W as in the engine."""
import os, torch
from loongx_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timed(d, it=20):
    for _ in range(3): ops.gemm([d])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): ops.gemm([d])
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
for bm, M, N in ((256, 4096, 4096), (128, 2048, 4096), (256, 8192, 4096)):
    os.environ["LX_GEMM_BM"] = str(bm)
    res = {}
    for K in (512, 1024, 2048, 3072, 6144, 12288):
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        d = ops.gemm_desc(A, W, C)
        res[K] = min(timed(d) for _ in range(3))
    ks = sorted(res)
    slope = (res[ks[-1]] - res[ks[2]]) / ((ks[-1] - ks[2]) / 64)
    icpt = res[ks[3]] - slope * ks[3] / 64
    rounds = (M // bm) * (N // 256) / 256
    print(f"BM={bm} M={M} N={N} ({rounds:.0f} round): " + "  ".join(f"K={k}: {res[k]:.1f}us" for k in ks) + f"  -> {slope/rounds:.3f} us per K tile per round, fixed {icpt:.1f} us; "
          f"loop rate {2*bm*256*64*256/(slope/rounds)/1e6:.0f} TF")
