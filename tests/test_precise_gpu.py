"""Precise mode (split-bf16 MFMA GEMMs, fp32 q/k/v, fp32 attention: loongx_amd/csrc/precise.hip + lx_gemm_split_kernel) -- the
arithmetic for the reference's shipped fp32 configuration (train/config/seed_512.yaml:2). Kernel parity vs torch fp32/fp64, the
engine vs the reference-generated goldens at fp32-class tolerance, and the north star's 1e-3 at full depth."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import load, relerr, tiny_transformer  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _pair(ops, x, lo_off=None, width=None):
    M, K = x.shape
    lo_off = K if lo_off is None else lo_off
    p = torch.zeros(M, width or (lo_off + K), dtype=torch.bfloat16, device=DEV)
    ops.split_bf16(x.contiguous(), p, lo_off)
    return p


def test_split_bf16_pair_carries_16_bits(ops):
    x = rnd(300, 512, seed=1)
    p = _pair(ops, x, lo_off=640, width=1200)
    hi, lo = p[:, :512].float(), p[:, 640:1152].float()
    assert torch.equal(hi, x.to(torch.bfloat16).float())
    assert torch.equal(lo, (x - hi).to(torch.bfloat16).float())
    assert relerr(hi + lo, x) < 2.0 ** -16


_WS = []


def _ws(ops):
    if not _WS:
        _WS.append(ops.gemm_workspace(DEV))
    return _WS[0]


@pytest.fixture
def gemm_plan(monkeypatch):
    """'w8': the 8-wave split kernels (no workspace); 'g4': lx_gemm4_kernel<true> forced, whole tiles only; 'g4sk': forced, with the
    workspace -- tails / short launches go through the two-workgroup split form. The library re-reads its switches afterwards."""
    from loongx_amd import _lib, ops

    def set_plan(plan):
        if plan == "w8":
            monkeypatch.setenv("LX_GEMM4", "0")
        else:
            monkeypatch.setenv("LX_GEMM4", "2")
            monkeypatch.setenv("LX_GEMM4_SK", "1" if plan == "g4sk" else "0")
        _lib.lib.lx_gemm_reload_env()
        return _ws(ops) if plan != "w8" else None
    yield set_plan
    for k in ("LX_GEMM4", "LX_GEMM4_SK"):
        os.environ.pop(k, None)
    monkeypatch.undo()
    _lib.lib.lx_gemm_reload_env()


@pytest.mark.parametrize("plan", ["w8", "g4", "g4sk"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 512, 192), (2560, 768, 3072), (2304, 7680, 1024)])
@pytest.mark.parametrize("segs", [2, 3])
def test_gemm_split_bf16(ops, M, N, K, segs, plan, gemm_plan):
    """k_segs = 2: A as a hi/lo pair x bf16-exact W; k_segs = 3: W = [W_hi | W_lo] too. fp32-class result either way, on the 8-wave
    split kernels and on lx_gemm4_kernel<true> (2560 x 768: 30 tiles = the split form's all-tiles case; 2304 x 7680: 270 tiles = one
    round + a 14-tile split tail)."""
    ws = gemm_plan(plan)
    A = rnd(M, K, seed=1)
    W = rnd(N, K, seed=2, scale=0.05)
    if segs == 2:
        W = W.to(torch.bfloat16).float()
    bias = rnd(N, seed=3)
    A2 = _pair(ops, A, lo_off=K + 64, width=2 * K + 128)
    hi = W.to(torch.bfloat16)
    Wd = hi if segs == 2 else torch.cat([hi, (W - hi.float()).to(torch.bfloat16)], 1).contiguous()
    ref = (A.double() @ W.double().T + bias.double()).float()
    for tiled in (False, True):
        if tiled and (N % 256 or K % 64):
            continue
        Wt = ops.tile_weight(Wd) if tiled else Wd
        C32 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm([ops.gemm_desc(A2, Wt, C32, bias=bias, epilogue=ops.LX_EPI_STORE_F32, K=K, N=N, k_segs=segs, a_lo_off=K + 64)], ws)
        assert relerr(C32, ref) < 3e-5, (segs, tiled)
    # hi/lo output pair with GELU
    Cp = torch.zeros(M, 2 * N + 64, dtype=torch.bfloat16, device=DEV)
    ops.gemm([ops.gemm_desc(A2, Wd, Cp, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, K=K, N=N, k_segs=segs,
                            a_lo_off=K + 64, c_lo_off=N + 64)], ws)
    want = torch.nn.functional.gelu(ref.double(), approximate="tanh").float()
    got = Cp[:, :N].float() + Cp[:, N + 64:2 * N + 64].float()
    assert relerr(got, want) < 5e-5
    assert relerr(Cp[:, :N].float(), want) < 4e-3                     # the hi image alone is the bf16 result
    if ws is not None:
        assert _status(ops, ws) == 0


def _status(ops, ws):
    from loongx_amd import _lib
    return _lib.lib.lx_gemm_workspace_status(ws.data_ptr(), torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("plan", ["w8", "g4", "g4sk"])
def test_gemm_split_bf16_lora_gate_residual(ops, plan, gemm_plan):
    """The precise mode's gated-residual launch with the LoRA term as cross-term slabs (engine._gemm_streams_p): two problems (one with
    the adapter, rank 4, three slabs), three K segments, 12 + 12 tiles, against fp64."""
    ws = gemm_plan(plan)
    M, N, K, r, B = 512, 768, 1024, 4, 2
    ref_all, probs, outs, keep = [], [], [], []
    for i in range(2):
        A = rnd(M, K, seed=10 + i)
        W = rnd(N, K, seed=20 + i, scale=0.03)
        bias = rnd(N, seed=30 + i)
        gate = rnd(B, N, seed=40 + i)
        X0 = rnd(M, N, seed=50 + i)
        A2 = _pair(ops, A, lo_off=K, width=2 * K)
        hi = W.to(torch.bfloat16)
        Wd = ops.tile_weight(torch.cat([hi, (W - hi.float()).to(torch.bfloat16)], 1).contiguous())
        y = A.double() @ W.double().T + bias.double()
        kw = {}
        if i == 1:
            down = rnd(r, K, seed=60, scale=0.05)
            up = rnd(N, r, seed=61, scale=0.05)
            t = A.double() @ down.double().T
            y = y + t @ up.double().T
            slabs = torch.zeros(3, M, r, dtype=torch.float32, device=DEV)        # the consumer adds the slabs: any decomposition of t
            slabs[0] = (t * 0.5).float(); slabs[1] = (t * 0.25).float(); slabs[2] = (t - slabs[0].double() - slabs[1].double()).float()
            kw = dict(lora_t=slabs[0], lora_up=up.contiguous(), lora_nsplit=3, lora_split_stride=slabs.stride(0))
            outs.append(slabs)
        C = X0.clone()
        g_rows = gate.double().repeat_interleave(M // B, 0)
        ref_all.append((X0.double() + g_rows * y).float())
        probs.append(ops.gemm_desc(A2, Wd, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, K=K, N=N, k_segs=3, a_lo_off=K, gate=gate,
                                   rows_per_batch=M // B, **kw))
        outs.append(C)
        keep.append((A2, Wd, bias, gate, kw))          # (a descriptor holds raw pointers: the operands must outlive the launch)
    ops.gemm(probs, ws)
    got = [o for o in outs if o.shape == (M, N)]
    for g, want in zip(got, ref_all):
        assert relerr(g, want) < 3e-5
    if ws is not None:
        assert _status(ops, ws) == 0


def test_ln_modulate_split(ops):
    M, D = 200, 3072
    X = rnd(M, D, seed=1, scale=2.0)
    sh, sc = rnd(2, D, seed=2, scale=0.3), rnd(2, D, seed=3, scale=0.3)
    Y = torch.zeros(M, 2 * D, dtype=torch.bfloat16, device=DEV)
    ops.ln_modulate_split_segs(X, [(0, M, 100, sh, sc)], Y, D, D)
    xd = X.double()
    ref = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-6)
    ref = ref * (1 + sc.double().repeat_interleave(100, 0)) + sh.double().repeat_interleave(100, 0)
    assert relerr(Y[:, :D].float() + Y[:, D:].float(), ref.float()) < 2e-5


def _attn_ref(q, k, v, lens, bias):
    """fp64 joint attention over segments with a (query segment, key segment) additive bias. q,k,v [B,H,S,128]."""
    S = sum(lens)
    m = torch.zeros(S, S, dtype=torch.float64, device=q.device)
    e = [0]
    for L in lens:
        e.append(e[-1] + L)
    for i in range(len(lens)):
        for j in range(len(lens)):
            m[e[i]:e[i + 1], e[j]:e[j + 1]] = bias[i][j]
    s = q.double() @ k.double().transpose(-1, -2) / math.sqrt(128.0) + m
    return torch.softmax(s, -1) @ v.double()


@pytest.mark.parametrize("lens,mode", [((16, 16, 16), "none"), ((40, 100, 70), "cfactor"), ((512, 1024, 1024), "none"),
                                       ((512, 1024, 1024), "nounion"), ((96, 200), "none")])
def test_attention_f32(ops, lens, mode):
    B, H = (2, 2) if sum(lens) < 1000 else (1, 3)
    D = H * 128
    ninf = float("-inf")
    bias = {"none": [[0.0] * 3] * 3, "cfactor": [[0, 0, math.log(0.5)], [0, 0, math.log(0.5)], [math.log(0.5), math.log(0.5), 0]],
            "nounion": [[0, 0, ninf], [0, 0, ninf], [ninf, ninf, 0]]}[mode]
    M = B * sum(lens)
    buf = rnd(M, 3 * D, seed=7)                                        # [k | v | q] like the engine's Y32
    O = torch.zeros(M, 2 * D + 64, dtype=torch.bfloat16, device=DEV)
    row0, r = [], 0
    for L in lens:
        row0.append(r)
        r += B * L
    ops.attn_fwd_f32(buf, O, q_col=2 * D, k_col=0, v_col=D, o_col=0, o_lo_off=D + 64, B=B, H=H, seg_row0=row0, seg_len=list(lens), bias=bias)
    got = O[:, :D].float() + O[:, D + 64:2 * D + 64].float()

    def gather(col):        # -> [B, H, S, 128]
        parts = [buf[row0[i]:row0[i] + B * L, col:col + D].view(B, L, H, 128) for i, L in enumerate(lens)]
        return torch.cat(parts, 1).permute(0, 2, 1, 3)
    ref = _attn_ref(gather(2 * D), gather(0), gather(D), lens, bias).permute(0, 2, 1, 3)          # [B, S, H, 128]
    e = 0
    for i, L in enumerate(lens):
        g = got[row0[i]:row0[i] + B * L].view(B, L, H, 128)
        assert relerr(g, ref[:, e:e + L].float()) < 2e-5, f"segment {i}"
        e += L


def test_qkv_prep_f32(ops):
    B, L, H = 2, 70, 2
    D = H * 128
    buf = rnd(B * L, 3 * D, seed=3)
    orig = buf.clone()
    wq, wk = rnd(128, seed=4).abs() + 0.5, rnd(128, seed=5).abs() + 0.5
    ids = torch.stack([torch.zeros(L), torch.arange(L).float() // 8, torch.arange(L).float() % 8], 1).to(DEV)
    cos, sin = ops.rope_table(ids)
    ops.qkv_prep_f32_segs(buf, 2 * D, 0, [(0, L, 0, wq, wk, cos, sin)], B, H)
    assert torch.equal(buf[:, D:2 * D], orig[:, D:2 * D])              # v untouched
    for col, w in ((2 * D, wq), (0, wk)):
        x = orig[:, col:col + D].view(B, L, H, 128).double()
        x = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * w.double()
        xr = torch.stack([-x[..., 1::2], x[..., 0::2]], -1).flatten(-2)
        ref = x * cos.double()[None, :, None, :] + xr * sin.double()[None, :, None, :]
        assert relerr(buf[:, col:col + D].view(B, L, H, 128), ref.float()) < 2e-6


def _split_images(ops, buf, B, H, lens, wq=None, wk=None, cos=None, sin=None):
    """[k | v | q] fp32 -> (QK2 [M, 4D] = [k_hi | k_lo | q_hi | q_lo], VT2 [2, B, H, 128, Spad], row0, vt0)"""
    D = H * 128
    row0, vt0, r, v = [], [], 0, 0
    for L in lens:
        row0.append(r); vt0.append(v)
        r += B * L
        v += (L + 63) // 64 * 64
    QK2 = torch.zeros(buf.shape[0], 4 * D, dtype=torch.bfloat16, device=DEV)
    VT2 = torch.zeros(2, B, H, 128, v, dtype=torch.bfloat16, device=DEV)
    segs = [(row0[i], L, vt0[i], wq, wk, cos[i] if cos else None, sin[i] if sin else None) for i, L in enumerate(lens)]
    before = buf.clone()
    ops.qkv_prep_split_segs(buf, 2 * D, 0, D, segs, B, H, QK2, q2_col=2 * D, k2_col=0, lo_off=D, VT2=VT2)
    assert torch.equal(buf, before)                                   # the fp32 projections are an input only
    return QK2, VT2, row0, vt0


def test_qkv_prep_split(ops):
    """RMSNorm + RoPE in fp32, then hi / lo pairs that carry 16 mantissa bits; V^T pair images in the attention kernel's layout."""
    B, L, H = 2, 70, 2
    D = H * 128
    buf = rnd(B * L, 3 * D, seed=3)
    wq, wk = rnd(128, seed=4).abs() + 0.5, rnd(128, seed=5).abs() + 0.5
    ids = torch.stack([torch.zeros(L), torch.arange(L).float() // 8, torch.arange(L).float() % 8], 1).to(DEV)
    cos, sin = ops.rope_table(ids)
    QK2, VT2, row0, vt0 = _split_images(ops, buf, B, H, [L], wq, wk, [cos], [sin])
    for col2, col, w in ((2 * D, 2 * D, wq), (0, 0, wk)):
        x = buf[:, col:col + D].view(B, L, H, 128).double()
        x = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * w.double()
        xr = torch.stack([-x[..., 1::2], x[..., 0::2]], -1).flatten(-2)
        ref = (x * cos.double()[None, :, None, :] + xr * sin.double()[None, :, None, :]).float().view(B * L, D)
        hi, lo = QK2[:, col2:col2 + D].float(), QK2[:, col2 + D:col2 + 2 * D].float()
        assert relerr(hi + lo, ref) < 5e-6                           # a bf16 pair carries 16 mantissa bits: <= 2^-17 per element, ~2.5e-6 rms
        assert torch.equal(lo, ((hi + lo) - hi).to(torch.bfloat16).float()) and relerr(hi, ref) > 1e-4      # a genuine pair, not hi alone
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    v = torch.zeros(B, 128, H, 128, device=DEV)
    v[:, :L] = buf[:, D:2 * D].view(B, L, H, 128)
    slots = (torch.arange(128) // 16) * 16 + perm[torch.arange(128) % 16]
    expect = v[:, slots.to(DEV)].permute(0, 2, 3, 1)                  # [B, H, d, slot]
    got = VT2[0].float() + VT2[1].float()
    assert relerr(got, expect) < 2.0 ** -16 and torch.equal(VT2[0].float(), expect.to(torch.bfloat16).float())


@pytest.mark.parametrize("lens,mode", [((16, 16, 16), "none"), ((40, 100, 70), "cfactor"), ((512, 1024, 1024), "none"),
                                       ((512, 1024, 1024), "nounion"), ((96, 200), "none"), ((300,), "none")])
def test_attention_split_bf16(ops, lens, mode):
    """lx_attn_fwd_split (hi.hi + hi.lo + lo.hi on the bf16 MFMA, fp32 softmax) against fp64 attention on the same fp32 q / k / v:
    fp32-class (the exact fp32-MFMA kernel's bound on the same cases is 2e-5), every mask mode, ragged segments."""
    B, H = (2, 2) if sum(lens) < 1000 else (1, 3)
    D = H * 128
    ninf = float("-inf")
    bias = {"none": [[0.0] * 3] * 3, "cfactor": [[0, 0, math.log(0.5)], [0, 0, math.log(0.5)], [math.log(0.5), math.log(0.5), 0]],
            "nounion": [[0, 0, ninf], [0, 0, ninf], [ninf, ninf, 0]]}[mode]
    M = B * sum(lens)
    buf = rnd(M, 3 * D, seed=7)                                        # [k | v | q]; no norm / RoPE (weights None): q, k pass through
    QK2, VT2, row0, vt0 = _split_images(ops, buf, B, H, list(lens))
    O = torch.zeros(M, 2 * D + 64, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd_split(QK2, VT2, O, q_col=2 * D, k_col=0, qk_lo_off=D, o_col=0, o_lo_off=D + 64, B=B, H=H, seg_row0=row0, seg_len=list(lens),
                       seg_vt0=vt0, bias=bias)
    got = O[:, :D].float() + O[:, D + 64:2 * D + 64].float()

    def gather(col):        # -> [B, H, S, 128]
        parts = [buf[row0[i]:row0[i] + B * L, col:col + D].view(B, L, H, 128) for i, L in enumerate(lens)]
        return torch.cat(parts, 1).permute(0, 2, 1, 3)
    ref = _attn_ref(gather(2 * D), gather(0), gather(D), lens, bias).permute(0, 2, 1, 3)          # [B, S, H, 128]
    e = 0
    for i, L in enumerate(lens):
        g = got[row0[i]:row0[i] + B * L].view(B, L, H, 128)
        assert relerr(g, ref[:, e:e + L].float()) < 3e-5, f"segment {i}"
        e += L
    again = torch.zeros_like(O)
    ops.attn_fwd_split(QK2, VT2, again, q_col=2 * D, k_col=0, qk_lo_off=D, o_col=0, o_lo_off=D + 64, B=B, H=H, seg_row0=row0,
                       seg_len=list(lens), seg_vt0=vt0, bias=bias)
    assert torch.equal(O, again)                                       # deterministic


@pytest.mark.parametrize("lens,mode", [((40, 100, 70), "cfactor"), ((512, 1024, 1024), "none"), ((512, 1024, 1024), "nounion"), ((300,), "none")])
def test_attention_split_bounded_scores(ops, lens, mode):
    """LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED on the split-bf16 kernel: q carries scale * log2 e (folded in fp32 before the hi / lo split),
    no running maximum, no rescale of O; the same fp32-class bound against fp64 attention as the max-tracking form."""
    B, H = (2, 2) if sum(lens) < 1000 else (1, 3)
    D = H * 128
    ninf = float("-inf")
    bias = {"none": [[0.0] * 3] * 3, "cfactor": [[0, 0, math.log(0.5)], [0, 0, math.log(0.5)], [math.log(0.5), math.log(0.5), 0]],
            "nounion": [[0, 0, ninf], [0, 0, ninf], [ninf, ninf, 0]]}[mode]
    M = B * sum(lens)
    buf = rnd(M, 3 * D, seed=13)
    buf[:, 2 * D:] *= 0.7                                              # |q.k| / sqrt(128) * log2 e stays far below the bound of 100
    scaled = buf.clone()
    scaled[:, 2 * D:] *= ops.Q_LOG2_FACTOR
    QK2, VT2, row0, vt0 = _split_images(ops, scaled, B, H, list(lens))
    O = torch.zeros(M, 2 * D, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd_split(QK2, VT2, O, q_col=2 * D, k_col=0, qk_lo_off=D, o_col=0, o_lo_off=D, B=B, H=H, seg_row0=row0, seg_len=list(lens),
                       seg_vt0=vt0, bias=bias, flags=ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED)
    got = O[:, :D].float() + O[:, D:].float()

    def gather(col):
        parts = [buf[row0[i]:row0[i] + B * L, col:col + D].view(B, L, H, 128) for i, L in enumerate(lens)]
        return torch.cat(parts, 1).permute(0, 2, 1, 3)
    ref = _attn_ref(gather(2 * D), gather(0), gather(D), lens, bias).permute(0, 2, 1, 3)
    e = 0
    for i, L in enumerate(lens):
        g = got[row0[i]:row0[i] + B * L].view(B, L, H, 128)
        assert relerr(g, ref[:, e:e + L].float()) < 3e-5, f"segment {i}"
        e += L


def test_attention_split_spiked_scores(ops):
    """An outlier key (raw q.k far above the row's other scores) late in the sequence: the running maximum jumps and every
    accumulator is rescaled; exact-max softmax must still agree with fp64."""
    B, H, L = 1, 2, 512
    D = H * 128
    buf = rnd(B * L, 3 * D, seed=11)
    buf[300, :128] = buf[17, 2 * D:2 * D + 128] * 6.0                 # key 300 of head 0 aligned with query 17
    QK2, VT2, row0, vt0 = _split_images(ops, buf, B, H, [L])
    O = torch.zeros(B * L, 2 * D, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd_split(QK2, VT2, O, q_col=2 * D, k_col=0, qk_lo_off=D, o_col=0, o_lo_off=D, B=B, H=H, seg_row0=row0, seg_len=[L], seg_vt0=vt0)
    got = (O[:, :D].float() + O[:, D:].float()).view(B, L, H, 128)
    q, k, v = (buf[:, c:c + D].view(B, L, H, 128).permute(0, 2, 1, 3) for c in (2 * D, 0, D))
    ref = _attn_ref(q, k, v, (L,), [[0.0] * 3] * 3).permute(0, 2, 1, 3)
    assert relerr(got, ref.float()) < 3e-5 and relerr(got[:, 17, 0], ref[:, 17, 0].float()) < 3e-5


# ---------------------------------------------------------------------------------------------------- engine
def _engine(tr, precise=True):
    from loongx_amd.flux.engine import DiTEngine
    from loongx_amd.flux.weights import FluxConfig, pack_state_dict
    c = tr.config
    cfg = FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                     attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                     pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)
    eng = DiTEngine(pack_state_dict(tr.state_dict(), cfg, "cuda", precise=precise), "cuda")
    eng.precise_default = precise
    return eng


def _run(eng, G, cond=True, c_t=0.0, guidance=True, model_config=None, c_factor=None):
    d = "cuda"
    eng.set_conditioning(G["in_enc"].to(d), G["in_pooled"].to(d), G["in_guidance"].to(d) if guidance else None, G["in_txt_ids"].to(d),
                         G["in_img_ids"].to(d), G["in_cond"].to(d) if cond else None, G["in_cond_ids"].to(d) if cond else None,
                         c_t=c_t, model_config=model_config or {}, c_factor=c_factor)
    return eng.forward(G["in_latents"].to(d), G["in_timestep"].to(d)).float().cpu().clone()


TOL_P = 2e-4      # fp32-class: the reference goldens are fp32 themselves


def test_precise_forward_matches_reference_goldens():
    """The tiny transformer's weights are NOT bf16-representable: this runs the 3-segment GEMMs and every residual pass."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    G = load("flux_tiny.npz")
    eng = _engine(tiny_transformer())
    assert any(k.endswith(".w2") for k in eng.w.t) and "mod.w_lo" in eng.w.t
    assert relerr(_run(eng, G), G["fwd_cond"]) < TOL_P
    assert relerr(_run(eng, G, cond=False), G["fwd_nocond"]) < TOL_P
    assert relerr(_run(eng, G, c_t=0.25), G["fwd_cond_ct025"]) < TOL_P
    eng2 = _engine(tiny_transformer(seed=3, guidance_embeds=False))
    assert relerr(_run(eng2, G, guidance=False), G["fwd_noguidance_seed3"]) < TOL_P
    # the same engine in bf16 mode (model_config overrides the default) lands at the bf16 tolerance, not the precise one
    e16 = relerr(_run(eng, G, model_config={"precise": False}), G["fwd_cond"])
    assert 5e-4 < e16 < 2.5e-2


@pytest.mark.parametrize("mc,cf", [({"union_cond_attn": False}, None), ({"independent_condition": True}, None), ({}, 0.5),
                                   ({"latent_lora": True}, None), ({"add_cond_attn": True}, None)])
def test_precise_model_config_modes(mc, cf):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import flux_ref as fr
    G = load("flux_tiny.npz")
    tr = tiny_transformer()
    if cf is not None:
        for m in tr.modules():
            if hasattr(m, "to_q"):
                m.c_factor = torch.ones(1, 1) * cf
    with torch.no_grad():
        want = fr.tranformer_forward(tr, G["in_cond"], G["in_cond_ids"], None, mc, hidden_states=G["in_latents"],
                                     encoder_hidden_states=G["in_enc"], pooled_projections=G["in_pooled"], timestep=G["in_timestep"],
                                     img_ids=G["in_img_ids"], txt_ids=G["in_txt_ids"], guidance=G["in_guidance"])[0]
    got = _run(_engine(tr), G, model_config=mc, c_factor=cf)
    assert relerr(got, want) < TOL_P


def test_precise_graph_replay_and_schedule_are_consistent():
    """Step graph replay == eager, and the prepared-schedule modulations == the per-step ones, in precise mode."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    G = load("flux_tiny.npz")
    eng = _engine(tiny_transformer())
    a = _run(eng, G)
    eng.use_graph = False
    b = _run(eng, G)
    assert torch.equal(a, b)


def test_full_depth_parity_precise():
    """The north star's bar: output parity to the fp32 reference path within 1e-3 rel-err -- at full depth (19 + 38 full-width
    blocks, S = 2560), every step of the 28-step trajectory and the final latents."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.parity import full_depth_parity
    rec = full_depth_parity("cuda:0", steps=28, every=3, precise=True)
    print("PARITY_PRECISE " + json.dumps(rec))
    assert rec["noise_pred_relerr_max"] < 1e-3, rec
    assert rec["final_latent_relerr"] < 1e-3, rec
    assert rec["final_latent_cosine"] > 0.999999, rec
