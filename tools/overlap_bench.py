"""Would running the MLP half of the single-block fused GEMM beside qkv_prep + attention (two graph branches) beat running
them one after the other? Serial on one stream vs concurrent on two streams, same kernels, same box."""
import torch
from loongx_amd import ops
dev, D, H = "cuda", 3072, 24
lens = (512, 1024, 1024); S = sum(lens)
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(S, D, device=dev, generator=g).to(torch.bfloat16)
Wm = ops.tile_weight((torch.randn(4 * D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16))
Wq = ops.tile_weight((torch.randn(3 * D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16))
Wf = ops.tile_weight((torch.randn(7 * D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16))
bias = torch.zeros(7 * D, device=dev)
Y = torch.zeros(S, 7 * D, device=dev, dtype=torch.bfloat16)
d_mlp = ops.gemm_desc(A, Wm, Y[:, 3 * D:], bias=bias[:4 * D], epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)
d_qkv = ops.gemm_desc(A, Wq, Y[:, :3 * D], bias=bias[:3 * D])
d_fused = ops.gemm_desc(A, Wf, Y, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, gelu_col_start=3 * D)
row0 = [0, lens[0], lens[0] + lens[1]]; vt0 = row0
VT = torch.zeros(1, H, 128, S, dtype=torch.bfloat16, device=dev)
segs = [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)]


def prep_attn():
    ops.qkv_prep_segs(Y, 2 * D, 0, D, segs, 1, H, VT)
    ops.attn_fwd(Y, Y, VT, Y, q_col=2 * D, k_col=0, o_col=2 * D, B=1, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0)


s2 = torch.cuda.Stream()


def serial_fused():
    ops.gemm([d_fused]); prep_attn()


def serial_split():
    ops.gemm([d_qkv]); ops.gemm([d_mlp]); prep_attn()


def concurrent():
    ops.gemm([d_qkv])
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        ops.gemm([d_mlp])
    prep_attn()
    torch.cuda.current_stream().wait_stream(s2)


def timed(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it


res = {}
for rep in range(3):
    for name, fn in (("fused GEMM -> prep -> attn", serial_fused), ("qkv -> mlp -> prep -> attn", serial_split), ("qkv -> [mlp || prep -> attn]", concurrent)):
        res.setdefault(name, []).append(timed(fn))
for k, v in res.items():
    print(f"{k:32s} {min(v):7.1f} us  (runs: {', '.join(f'{x:.1f}' for x in v)})")
