"""Phase times of lx_gemm4_kernel's split form per workgroup (a -DLX_G4_PROBE build: LX_AMD_LIB=loongx_amd/lib/liblx_amd_g4probe.so):
M = 2560, N = 3072 (120 tiles -> 240 workgroups: even = part 0 (parks blocks 4-7, owns 0-3), odd = part 1), gated fp32 residual epilogue. PK = K.
stamps: 0 start | 1 K tile 0 landed | 2 main loop starts | 3 main loop done | 4 DMA landed + barrier | 6 own blocks parked, partner's flag seen | 5 done."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["LX_GEMM4"] = "2"; os.environ["LX_GEMM_PAIR_MIN_KT"] = "16"
from loongx_amd import ops, _lib
dev = "cuda"
M, N, K = 2560, 3072, int(os.environ.get("PK", "3072"))
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
W = ops.tile_weight((torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
C = torch.randn(M, N, device=dev, generator=g)
gate = torch.randn(1, N, device=dev, generator=g)
bias = torch.randn(N, device=dev, generator=g)
ws = ops.gemm_workspace(dev)
d = ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=M)
for _ in range(5): ops.gemm([d], ws)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.gemm([d], ws)
e.record(); torch.cuda.synchronize()
n = 240 * 8
host = (ctypes.c_ulonglong * n)()
_lib.lib.lx_g4_probe_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_size_t]
assert _lib.lib.lx_g4_probe_read(host, n) == 0
t = torch.tensor(list(host), dtype=torch.float64).view(240, 8) * 0.01      # us
t0 = t[:, 0].min()
own, park = t[0::2], t[1::2]
print(f"K = {K}: launch {s.elapsed_time(e) * 1e3 / 20:.1f} us (probe build); stamps relative to the launch's first workgroup start, mean (min, max) us")
def row(nm, x): print(f"  {nm:44s} {float(x.mean()):7.2f} ({float(x.min()):6.2f}, {float(x.max()):6.2f})")
row("part 0: start -> K tile 0 landed [1-0]", own[:, 1] - own[:, 0])
row("part 0: main loop [3-2]", own[:, 3] - own[:, 2])
row("part 1: main loop [3-2]", park[:, 3] - park[:, 2])
row("part 0: drain + barrier [4-3]", own[:, 4] - own[:, 3])
row("part 0: park blocks 4-7 + flags [6-4]", own[:, 6] - own[:, 4])
row("part 1: park blocks 0-3 + flags [6-4]", park[:, 6] - park[:, 4])
row("part 0: epilogue of blocks 0-3 [5-6]", own[:, 5] - own[:, 6])
row("part 1: epilogue of blocks 4-7 [5-6]", park[:, 5] - park[:, 6])
row("part 0: whole life [5-0]", own[:, 5] - own[:, 0])
row("part 1: whole life [5-0]", park[:, 5] - park[:, 0])
print("  (stamps are s_memtime ticks x 0.01; calibrate against the main loop: K / 128 K tiles per half)")

# reference: the same gated-residual epilogue on WHOLE tiles (M = 4096, N = 4096: 256 tiles, one round, no exchange)
M2 = N2 = 4096
A2 = torch.randn(M2, K, device=dev, generator=g).to(torch.bfloat16)
W2 = ops.tile_weight((torch.randn(N2, K, device=dev, generator=g) * 0.02).to(torch.bfloat16))
C2 = torch.randn(M2, N2, device=dev, generator=g)
gate2 = torch.randn(1, N2, device=dev, generator=g); bias2 = torch.randn(N2, device=dev, generator=g)
d2 = ops.gemm_desc(A2, W2, C2, bias=bias2, epilogue=ops.LX_EPI_RESID_F32, gate=gate2, rows_per_batch=M2)
for _ in range(5): ops.gemm([d2], ws)
torch.cuda.synchronize()
n2 = 256 * 8
host2 = (ctypes.c_ulonglong * n2)()
assert _lib.lib.lx_g4_probe_read(host2, n2) == 0
t2 = torch.tensor(list(host2), dtype=torch.float64).view(256, 8) * 0.01
row("whole tiles: start -> first DMA piece issued [7-0]", t2[:, 7] - t2[:, 0])
row("whole tiles: start -> K tile 0 landed [1-0]", t2[:, 1] - t2[:, 0])
row("whole tiles: fragment reads [2-1]", t2[:, 2] - t2[:, 1])
row("whole tiles: main loop [3-2]", t2[:, 3] - t2[:, 2])
row("whole tiles: drain + barrier [4-3]", t2[:, 4] - t2[:, 3])
row("whole tiles: gated-residual epilogue [5-4]", t2[:, 5] - t2[:, 4])
row("whole tiles: whole life [5-0]", t2[:, 5] - t2[:, 0])
# and the bf16 + GELU store epilogue (ff1's), same tiles
C3 = torch.empty(M2, N2, device=dev, dtype=torch.bfloat16)
d3 = ops.gemm_desc(A2, W2, C3, bias=bias2, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)
for _ in range(5): ops.gemm([d3], ws)
torch.cuda.synchronize()
assert _lib.lib.lx_g4_probe_read(host2, n2) == 0
t3 = torch.tensor(list(host2), dtype=torch.float64).view(256, 8) * 0.01
row("whole tiles, bf16 + GELU: start -> K tile 0 landed [1-0]", t3[:, 1] - t3[:, 0])
row("whole tiles, bf16 + GELU: main loop [3-2]", t3[:, 3] - t3[:, 2])
row("whole tiles, bf16 + GELU: epilogue [5-4]", t3[:, 5] - t3[:, 4])
row("whole tiles, bf16 + GELU: whole life [5-0]", t3[:, 5] - t3[:, 0])
