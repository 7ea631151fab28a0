"""K sweep on the step's own output shapes (planner's choice of tiles): per-K-tile cost per CU-round-equivalent vs the 1.50 us of a
cache-resident square problem (tools/gemm_slope.py)."""
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
import sys
SHAPES = ((2560, 21504, (1024, 2048, 3072, 6144)), (2560, 12288, (1024, 2048, 3072, 6144)), (2560, 9216, (1024, 2048, 3072, 6144)), (2560, 3072, (3072, 6144, 12288, 15360)),
          (4096, 16384, (1024, 2048, 3072, 6144)), (16384, 4096, (1024, 2048, 3072, 6144)))
if len(sys.argv) > 1: SHAPES = tuple(SHAPES[int(i)] for i in sys.argv[1].split(","))
for M, N, Ks in SHAPES:
    res = {}
    for K in Ks:
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res[K] = min(timed(ops.gemm_desc(A, W, C)) for _ in range(3))
    ks = sorted(res)
    slope = (res[ks[-1]] - res[ks[0]]) / ((ks[-1] - ks[0]) / 64)
    units = M * N / (256 * 256) / 256        # 256x256-tile rounds of work
    print(f"M={M} N={N}: " + "  ".join(f"K={k}: {res[k]:.1f}us" for k in ks) + f"  -> {slope/units:.3f} us per K tile per round-equivalent ({units:.2f} rounds), "
          f"fixed {res[ks[0]] - slope*ks[0]/64:.1f} us; {2*M*N*64/slope/1e6:.0f} TF marginal")
