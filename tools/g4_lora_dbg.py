import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from loongx_amd import ops
dev = "cuda"; torch.manual_seed(0)
M, K, r, N = 512, 1024, 4, 512
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
Ad = (torch.randn(r, K, device=dev) * 0.1).to(torch.bfloat16); Bu = torch.randn(N, r, device=dev) * 0.2
t = torch.zeros(M, 16, device=dev); ops.lora_down(A, Ad, t)
C = torch.zeros(M, N, device=dev)
ops.gemm([ops.gemm_desc(A, W, C, epilogue=ops.LX_EPI_STORE_F32, lora_t=t, lora_up=Bu)])
ref = A.float() @ W.float().T + t[:, :r] @ Bu.T
base = A.float() @ W.float().T
err = (C - ref).abs()
print("relerr", float((C - ref).norm() / ref.norm()), "without lora term", float((C - base).norm() / ref.norm()))
e = err.view(M // 16, 16, N // 16, 16).amax((1, 3))
print("16x16 blocks with error > 1e-3:", int((e > 1e-3).sum()), "of", e.numel())
bad = (e > 1e-3).nonzero()
print(bad[:20].tolist())
el = err.view(M // 16, 16, N // 16, 16)[0, :, 0, :]
print((el > 1e-3).int())
