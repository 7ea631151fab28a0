"""Reference point, not a product path: the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, Tensile assembly kernels) on the denoise
step's own shapes beside lx_gemm_bf16 with its plain bf16-store epilogue, same random operands, same box, alternating, sustained loops
(the part is power-capped: 200 back-to-back launches per sample). Prints us per launch and TFLOP/s for both."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from loongx_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = ((2560, 21504, 3072, "single fused [k|v|q|mlp]"), (2560, 12288, 3072, "double ff1"), (2560, 9216, 3072, "double q/k/v (one weight)"),
          (2560, 3072, 12288, "ff2"), (2560, 3072, 15360, "single proj_out"), (2560, 3072, 3072, "to_out"),
          (40960, 21504, 3072, "single fused, batch 16"), (8192, 8192, 8192, "8k cube"))
IT = int(os.environ.get("GV_IT", "200"))
def timed(fn, it):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
for M, N, K, name in SHAPES:
    it = max(20, min(IT, int(IT * 2560 * 21504 * 3072 / (M * N * K))))
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    Wr = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    Wt = ops.tile_weight(Wr.clone())
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = ops.gemm_desc(A, Wt, C)
    WrT = Wr.t()
    ws = ops.gemm_workspace(dev) if os.environ.get("GV_WS", "1") != "0" else None      # (the engine's default plans pass the workspace)
    lx = lambda: ops.gemm([d], workspace=ws)
    vendor = lambda: torch.matmul(A, WrT, out=C2)
    best = {"lx": 1e9, "vendor": 1e9}
    for _ in range(3):
        best["lx"] = min(best["lx"], timed(lx, it))
        best["vendor"] = min(best["vendor"], timed(vendor, it))
    err = float((C.float() - C2.float()).norm() / C2.float().norm())
    fl = 2.0 * M * N * K
    print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d}: lx {best['lx']:8.1f} us {fl / best['lx'] / 1e6:6.0f} TF | vendor {best['vendor']:8.1f} us {fl / best['vendor'] / 1e6:6.0f} TF | "
          f"lx / vendor time {best['lx'] / best['vendor']:.3f}  (outputs differ by {err:.1e})", flush=True)
