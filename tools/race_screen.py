"""Race screen for the hand-scheduled kernels: (1) every hot kernel run many times on identical inputs must give bit-identical
outputs (a DMA / LDS ordering race shows up as run-to-run differences), (2) randomised ragged attention shapes vs fp32 SDPA."""
import math, random, sys
import torch
from loongx_amd import ops
sys.path.insert(0, ".")
from tests.test_kernels_gpu import _attn_reference, _qkv_buffer, _segments, BIASES
from tests.helpers import relerr
dev = "cuda"
torch.manual_seed(0)
bad = 0

def same(name, fn, n=60):
    global bad
    ref = fn().clone()
    for i in range(n):
        if not torch.equal(fn(), ref):
            print("NONDETERMINISTIC:", name, "run", i); bad += 1; return
    print("ok  ", name)

# ---- GEMMs: every plan (256 / 128 / mixed one-grid), epilogues, cold weights rotating
D = 3072
for (M, N, K, resid) in ((2560, 3 * D, D, False), (2560, 7 * D, D, False), (2560, D, D, True), (2560, D, 4 * D, True), (1000, 768, 256, False)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
    bias = torch.randn(N, device=dev)
    if resid:
        X0 = torch.randn(M, N, device=dev); gate = torch.randn(1, N, device=dev)
        def fn(A=A, W=Ws[0], X0=X0, gate=gate, bias=bias):
            C = X0.clone(); ops.gemm([ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate)]); return C
    else:
        def fn(A=A, W=Ws[0], bias=bias, M=M, N=N):
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); ops.gemm([ops.gemm_desc(A, W, C, bias=bias)]); return C
    same(f"gemm M={M} N={N} K={K} resid={resid}", fn, n=40)

# ---- projection GEMM with the LX_EPI_QKV epilogue (cross-wave sums through LDS, inline-asm table loads with counted vmcnt)
for Bq in (1, 4):
    Hq, Dq = 24, 3072
    lens_q = (512, 1024, 1024)
    Mq = Bq * sum(lens_q)
    Aq = torch.randn(Mq, Dq, device=dev).to(torch.bfloat16)
    Wq = (torch.randn(7 * Dq, Dq, device=dev) * 0.02).to(torch.bfloat16)
    bq_ = torch.randn(7 * Dq, device=dev) * 0.1
    wn = 1 + 0.1 * torch.randn(128, device=dev)
    ropes = []
    for L_ in lens_q:
        ang = torch.rand(L_, 64, device=dev) * 6.28
        cs = torch.empty(L_, 128, device=dev); cs[:, 0::2] = ang.cos(); cs[:, 1::2] = ang.sin()
        ropes.append(cs)
    r0, v0, r_, p_ = [], [], 0, 0
    for L_ in lens_q:
        r0.append(r_); v0.append(p_); r_ += Bq * L_; p_ += (L_ + 63) // 64 * 64
    VTq = torch.zeros(Bq, Hq, 128, p_, dtype=torch.bfloat16, device=dev)
    Cq = torch.zeros(Mq, 7 * Dq, dtype=torch.bfloat16, device=dev)
    def fnq(Bq=Bq, Aq=Aq, Wq=Wq, bq_=bq_, Cq=Cq, VTq=VTq, r0=r0, v0=v0, ropes=ropes):
        probs = []
        for i, L_ in enumerate(lens_q):
            rows = slice(r0[i], r0[i] + Bq * L_)
            probs.append(ops.gemm_desc(Aq[rows], Wq, Cq[rows], bias=bq_, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, rows_per_batch=L_,
                                       gelu_col_start=3 * Dq, qkv=dict(norm_q=wn, norm_k=wn, rope=ropes[i], vt=VTq, vt_pos0=v0[i], d=Dq)))
        ops.gemm(probs)
        return torch.cat([Cq.flatten().view(torch.int16).float(), VTq.flatten().view(torch.int16).float()])
    same(f"gemm LX_EPI_QKV fused single-block projection B={Bq}", fnq, n=40)

# ---- attention (bf16 pipelined, fp8) at the model shape
B, H = 1, 24
lens = (512, 1024, 1024); Dm = H * 128
buf = _qkv_buffer(B, lens, H, seed=3)
row0, vt0, vt_len = _segments(B, lens)
VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=dev)
ops.qkv_prep_segs(buf, 2 * Dm, 0, Dm, [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)], B, H, VT)
def attn():
    O = torch.empty(buf.shape[0], Dm, dtype=torch.bfloat16, device=dev)
    ops.attn_fwd(buf, buf, VT, O, q_col=2 * Dm, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0); return O
same("attention bf16 (pipelined) S=2560 H=24", attn, n=100)
Q8 = torch.zeros(buf.shape[0], Dm, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8); VT8 = torch.zeros(B, H, 128, vt_len, dtype=torch.uint8, device=dev)
ops.qkv_prep_fp8_segs(buf, 2 * Dm, 0, Dm, [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)], B, H, Q8, K8, VT8)
def attn8():
    O = torch.empty(buf.shape[0], Dm, dtype=torch.bfloat16, device=dev)
    ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0); return O
same("attention fp8 S=2560 H=24", attn8, n=100)

# ---- randomised ragged attention vs fp32 reference
rng = random.Random(1)
worst = 0.0
for trial in range(24):
    nseg = rng.choice((1, 2, 3)); ls = tuple(rng.choice((1, 7, 63, 64, 65, 130, 257, 500, 700)) for _ in range(nseg)); Bb, Hh = rng.choice((1, 2)), rng.choice((1, 3))
    mode = rng.choice(list(BIASES)) if nseg == 3 else "none"
    bias = [row[:] for row in BIASES[mode]]
    b2 = _qkv_buffer(Bb, ls, Hh, seed=100 + trial); o2 = b2.clone(); Dh = Hh * 128
    r0, v0, vl = _segments(Bb, ls)
    VT2 = torch.zeros(Bb, Hh, 128, vl, dtype=torch.bfloat16, device=dev)
    ops.qkv_prep_segs(b2, 2 * Dh, 0, Dh, [(r0[i], ls[i], v0[i], None, None, None, None) for i in range(nseg)], Bb, Hh, VT2)
    ops.attn_fwd(b2, b2, VT2, b2, q_col=2 * Dh, k_col=0, o_col=2 * Dh, B=Bb, H=Hh, seg_row0=r0, seg_len=list(ls), seg_vt0=v0, bias=bias)
    ref, edges = _attn_reference(o2, Bb, Hh, ls, bias, 2 * Dh, 0, Dh)
    got = b2.float().cpu()
    for s_, L_ in enumerate(ls):
        e = float(relerr(got[r0[s_]: r0[s_] + Bb * L_, 2 * Dh: 3 * Dh].view(Bb, L_, Hh, 128), ref[:, edges[s_]:edges[s_ + 1]]))
        worst = max(worst, e)
        if e > 6e-3: print("MISMATCH", ls, mode, s_, e); bad += 1
print(f"random ragged attention: worst rel err {worst:.2e} over 24 shape sets")
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FOUND {bad} PROBLEMS")
