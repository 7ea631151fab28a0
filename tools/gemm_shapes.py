"""The six GEMM launches of a denoise step in the form the engine issues them (3 row groups, pre-tiled W, bias, LoRA on the condition
rows, GELU / gated fp32 residual epilogues), each timed back to back (min of 3 passes), and their per-step total.
A/B two builds in one gpurun call with LX_AMD_LIB=<other .so>; env knobs of the library apply (LX_GEMM_PERSISTENT, ...)."""
import os, sys
import torch
from loongx_amd import ops
dev, D, r = "cuda", 3072, 4
Ms = [512, 1024, 1024]; S = sum(Ms)
g = torch.Generator(device=dev).manual_seed(0)
def rn(*s, scale=1.0, dt=torch.float32): return (torch.randn(*s, device=dev, generator=g) * scale).to(dt)
def launch(N, K, kind, nmod):
    A = rn(S, K, dt=torch.bfloat16); W = ops.tile_weight(rn(N, K, scale=0.02, dt=torch.bfloat16)); Wt = ops.tile_weight(rn(N, K, scale=0.02, dt=torch.bfloat16))
    bias = rn(N, scale=0.1); Ad = rn(nmod * r, K, scale=0.1, dt=torch.bfloat16); Bu = rn(N, r, scale=0.1)
    Tls = torch.zeros(4, Ms[2], 16, dtype=torch.float32, device=dev); Tl = Tls[0]          # 4 K-split slabs, as the engine uses
    ops.lora_down(A[1536:], Ad, Tl[:, :nmod * r], n_split=4, split_stride=Tls.stride(0))
    gate = rn(3, N)
    C = rn(S, N) if kind == "resid" else torch.empty(S, N, device=dev, dtype=torch.bfloat16)
    ds, r0 = [], 0
    for i, M in enumerate(Ms):
        kw = dict(bias=bias, rows_per_batch=M)
        if kind == "resid": kw.update(epilogue=ops.LX_EPI_RESID_F32, gate=gate[i:i + 1])
        elif kind == "gelu": kw.update(epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)
        elif kind == "fused": kw.update(epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, gelu_col_start=3 * D)
        if i == 2 and not os.environ.get("NOLORA"):
            kw.update(lora_t=Tl, lora_up=Bu, lora_mod_cols=D if nmod > 1 else 0, lora_toff_max=nmod - 1, lora_nsplit=4, lora_split_stride=Tls.stride(0))
        ds.append(ops.gemm_desc(A[r0:r0 + M], Wt if (i == 0 and kind != "fused" and N != D * 0) else W, C[r0:r0 + M], **kw)); r0 += M
    return ds, (A, W, Wt, bias, Ad, Bu, Tls, gate, C)
WS = ops.gemm_workspace(dev) if os.environ.get('LX_PAIR_PLAN', '1') != '0' else None      # the engine's per-stream pair-plan workspace
def timed(ds, it=20):
    for _ in range(3): ops.gemm(ds, WS)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): ops.gemm(ds, WS)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
specs = [("qkv", 3 * D, D, "bf16", 3, 19), ("attn out", D, D, "resid", 1, 19), ("ff1", 4 * D, D, "gelu", 1, 19), ("ff2", D, 4 * D, "resid", 1, 19),
         ("single fused", 7 * D, D, "fused", 4, 38), ("single out", D, 5 * D, "resid", 1, 38)]
built = [(n, launch(N, K, kind, nmod), cnt, 2.0 * S * N * K) for n, N, K, kind, nmod, cnt in specs]
best = {n: 1e9 for n, *_ in built}
for _ in range(3):
    for n, (ds, keep), cnt, fl in built: best[n] = min(best[n], timed(ds))
tot = sum(best[n] * cnt for n, _, cnt, _ in built)
print(os.environ.get("TAG", ""), " ".join(f"{n}: {best[n]:.1f}us ({fl/best[n]/1e6:.0f}TF)" for n, _, cnt, fl in built), f"| per step {tot/1e3:.2f} ms")
