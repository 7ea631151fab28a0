"""Phase times of lx_gemm4_kernel per workgroup (a -DLX_G4_PROBE build: LX_AMD_LIB=loongx_amd/lib/liblx_amd_g4probe.so):
0 start | 1 K tile 0 landed | 2 main loop starts | 3 main loop done | 4 all DMA landed + barrier | 5 epilogue stores done. 100 MHz ticks."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["LX_GEMM4"] = "2"
from loongx_amd import ops, _lib
dev = "cuda"
M, N, K = 4096, 16384, int(os.environ.get("PK", "3072"))
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
d = ops.gemm_desc(A, W, C)
for _ in range(5): ops.gemm([d])
torch.cuda.synchronize()
n = 1024 * 8
host = (ctypes.c_ulonglong * n)()
_lib.lib.lx_g4_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_size_t]
assert _lib.lib.lx_g4_probe_read(host, n) == 0
t = torch.tensor(list(host), dtype=torch.float64).view(1024, 8)
d_ = (t[:, 1:6] - t[:, 0:5]) * 10.0 / 1000.0        # us
names = ["start -> K tile 0 landed", "fragment reads (+LoRA)", "main loop", "drain + barrier", "epilogue"]
print(f"K = {K}: per workgroup, mean over 1024 tiles (us)")
for i, nm in enumerate(names): print(f"  {nm:28s} {float(d_[:, i].mean()):7.2f}   (min {float(d_[:, i].min()):6.2f}, max {float(d_[:, i].max()):6.2f})")
print(f"  total                        {float((t[:, 5] - t[:, 0]).mean()) * 0.01:7.2f}; launch span {float(t[:, 5].max() - t[:, 0].min()) * 0.01:7.1f} us for 4 rounds")
