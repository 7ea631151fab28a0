"""fp8 (e4m3) attention path (BASELINE configs[4]: "fp8 MFMA attention path") against fp32 references.

Tolerance (stated here, fp8-specific): e4m3 keeps 3 mantissa bits, i.e. a relative rounding error uniform in +-2^-4 (3.6 % rms)
per element. V and the probabilities P are each rounded once; on the RANDOM, uncorrelated V of these tests the output
sum_j p_j v_j and its rounding noise both shrink like sqrt(sum p_j^2), so the relative L2 error is sqrt(2) * 3.6 % = 5.1 % plus
the score perturbation from rounding q and k, independent of the sequence length: measured 4.5e-2 .. 5.4e-2 on every case
(16 to 8704 keys). The asserts allow 7e-2; the bf16 path's bound on the same tests is 6e-3."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import relerr  # noqa: E402
from tests.test_kernels_gpu import BIASES, DEV, _attn_reference, _qkv_buffer, _segments, ops, rnd  # noqa: E402,F401

TOL_FP8 = 7e-2


def _run_fp8(ops, buf, B, H, lens, bias, wq=None, wk=None, cos=None, sin=None):
    D = H * 128
    row0, vt0, vt_len = _segments(B, lens)
    M = buf.shape[0]
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=DEV)
    K8 = torch.zeros(M, D, dtype=torch.uint8, device=DEV)
    VT8 = torch.zeros(B, H, 128, vt_len, dtype=torch.uint8, device=DEV)
    segs = [(row0[s], lens[s], vt0[s], wq, wk, cos[s] if cos else None, sin[s] if sin else None) for s in range(len(lens))]
    before = buf.clone()
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
    assert torch.equal(buf, before)                                  # the bf16 buffer is an input only
    O = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=bias)
    return O, (Q8, K8, VT8), (row0, vt0)


@pytest.mark.parametrize("mode", list(BIASES))
@pytest.mark.parametrize("lens", [(16, 16, 16), (64, 128, 200), (512, 1024, 1024)])
def test_fp8_attention_segments(ops, lens, mode):
    B, H = (1, 2) if lens[0] == 512 else (2, 3)
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=5)
    bias = BIASES[mode]
    O, _, (row0, _) = _run_fp8(ops, buf, B, H, lens, bias)
    ref, edges = _attn_reference(buf, B, H, lens, bias, 2 * D, 0, D)
    got = O.float().cpu()
    for s, Ls in enumerate(lens):
        o = got[row0[s]: row0[s] + B * Ls].view(B, Ls, H, 128)
        assert relerr(o, ref[:, edges[s]:edges[s + 1]]) < TOL_FP8, f"segment {s}"


@pytest.mark.parametrize("peaked", [False, True])
def test_fp8_attention_probability_forms(ops, peaked):
    """Round 6: the default probability bytes are the log-linear code of the score (one v_cvt_pk_u8_f32 per score; include/lx.h LX_ATTN_P_EXP2
    is the v_exp_f32 + e4m3-rounding form), and both carry the constant factor 2^6 that moves e4m3's flush-to-zero from 2^-9.5 to 2^-15.5 of the
    running reference. Both forms hold the fp8 tolerance against fp32 attention and agree with each other within the e4m3 step; on PEAKED rows
    (a few keys 2x the norm of the rest, 2560 keys: the far tail carries real mass) the exponential form must stay within the same tolerance
    -- with the reference at 2^0 (round 5) that case measured 1.4 - 2.5x the flat-row error (tools/p_loglin_emulation.py)."""
    B, H, lens = 1, 2, (512, 1024, 1024)
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=11)
    if peaked:
        buf[::97, :D] *= 2.0                                      # k columns of every 97th row
    bias = BIASES["none"]
    row0, vt0, vt_len = _segments(B, lens)
    M = buf.shape[0]
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=DEV); K8 = torch.zeros_like(Q8)
    VT8 = torch.zeros(B, H, 128, vt_len, dtype=torch.uint8, device=DEV)
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, [(row0[s], lens[s], vt0[s], None, None, None, None) for s in range(3)], B, H, Q8, K8, VT8)
    ref, edges = _attn_reference(buf, B, H, lens, bias, 2 * D, 0, D)
    outs = {}
    for name, fl in (("loglin", 0), ("exp2", ops.ATTN_P_EXP2)):
        O = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
        ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=bias, flags=fl)
        outs[name] = torch.cat([O.float().cpu()[row0[s]: row0[s] + B * lens[s]].view(B, lens[s], H, 128) for s in range(3)], 1)
    errs = {n: relerr(o, ref) for n, o in outs.items()}
    assert errs["exp2"] < TOL_FP8 and errs["loglin"] < 1.35 * TOL_FP8, errs      # (the chord of 2^f: up to +26 % on peaked rows, tools/attn_fp8_ab.py)
    assert errs["loglin"] < 1.4 * errs["exp2"] + 1e-3, errs
    assert relerr(outs["loglin"], outs["exp2"]) < TOL_FP8, errs


def test_fp8_images_hold_the_normalised_rotated_values(ops):
    """Q8 / K8 = e4m3(16 * RoPE(RMSNorm(x) * w)); VT8 = e4m3(v) in the MFMA key order."""
    from oracle.flux_modules import apply_rotary_emb, rope_tables
    B, H, Ls = 2, 3, 70
    D = H * 128
    buf = _qkv_buffer(B, [Ls], H, seed=1)
    wq, wk = 1 + 0.1 * rnd(128, seed=2), 1 + 0.1 * rnd(128, seed=3)
    ids = torch.zeros(Ls, 3)
    ids[:, 1] = torch.arange(Ls) // 8
    ids[:, 2] = torch.arange(Ls) % 8 - 3
    cos, sin = rope_tables(ids)
    _, (Q8, K8, VT8), _ = _run_fp8(ops, buf, B, H, (Ls,), BIASES["none"], wq=wq, wk=wk, cos=[cos.to(DEV)], sin=[sin.to(DEV)])
    o = buf.float().cpu().view(B, Ls, 3, H, 128)

    def ref(x, w):
        x = x.permute(0, 2, 1, 3)
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.cpu()
        return apply_rotary_emb(x, (cos, sin)).permute(0, 2, 1, 3)

    q8 = Q8.view(torch.float8_e4m3fn).float().cpu().view(B, Ls, H, 128) / ops.FP8_Q_SCALE
    k8 = K8.view(torch.float8_e4m3fn).float().cpu().view(B, Ls, H, 128) / ops.FP8_K_SCALE
    assert relerr(q8, ref(o[:, :, 2], wq)) < 4e-2 and relerr(k8, ref(o[:, :, 0], wk)) < 4e-2       # one e4m3 rounding per element
    # V^T image: byte j = g*32 + p of a 64-key tile row holds key (p>>4)*32 + 8*((p&15)>>2) + 4*g + (p&3)
    j = torch.arange(64)
    g, p = j >> 5, j & 31
    key = (p >> 4) * 32 + 8 * ((p & 15) >> 2) + 4 * g + (p & 3)
    v = torch.zeros(B, 128, H, 128)
    v[:, :Ls] = o[:, :, 1]
    slots = torch.cat([key, 64 + key])                                # two tiles cover the 70 keys (padded to 128)
    expect = v[:, slots].permute(0, 2, 3, 1).to(torch.float8_e4m3fn).float()
    assert torch.equal(VT8.view(torch.float8_e4m3fn).float().cpu(), expect)


def test_fp8_attention_at_1024sq_sequence_length(ops):
    lens, B, H = (512, 4096, 4096), 1, 2
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=21)
    O, _, (row0, _) = _run_fp8(ops, buf, B, H, lens, BIASES["cfactor"])
    ref, edges = _attn_reference(buf, B, H, lens, BIASES["cfactor"], 2 * D, 0, D)
    got = O.float().cpu()
    for s, Ls in enumerate(lens):
        assert relerr(got[row0[s]: row0[s] + B * Ls].view(B, Ls, H, 128), ref[:, edges[s]:edges[s + 1]]) < TOL_FP8, f"segment {s}"


def test_engine_fp8_attention_vs_bf16_path_at_1024sq():
    """configs[4] shape end to end through the engine (full width, 1 + 1 blocks, 512 + 4096 + 4096 tokens): the velocity with
    model_config["attn_fp8"] stays within the fp8 tolerance of the bf16 path, is deterministic and batch-independent."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import flux_modules as fm
    from loongx_amd.flux.engine import DiTEngine
    from loongx_amd.flux.weights import FluxConfig, synthetic_weights
    eng = DiTEngine(synthetic_weights(FluxConfig(num_layers=1, num_single_layers=1), "cuda", seed=0), "cuda")
    hw, T = 64, 512
    N = hw * hw
    ids = fm.prepare_latent_image_ids(hw, hw).cuda()
    cids = ids.clone()
    cids[:, 2] -= hw
    g = torch.Generator(device="cuda").manual_seed(23)
    B = 2
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = dict(lat=r(B, N, 64), cond=r(B, N, 64), pe=r(B, T, 4096) * 0.1, pooled=r(B, 768))

    def fwd(sl, mc):
        n = x["lat"][sl].shape[0]
        eng.set_conditioning(x["pe"][sl], x["pooled"][sl], torch.full((n,), 3.5, device="cuda"), torch.zeros(T, 3, device="cuda"), ids,
                             x["cond"][sl], cids, model_config=mc)
        return eng.forward(x["lat"][sl], torch.full((n,), 0.7, device="cuda")).clone()

    ref = fwd(slice(None), {})
    v8 = fwd(slice(None), {"attn_fp8": True})
    assert torch.isfinite(v8).all()
    e = relerr(v8.cpu(), ref.cpu())
    assert 1e-4 < e < TOL_FP8, e                                         # the fp8 path is live and within its tolerance
    assert torch.equal(v8, fwd(slice(None), {"attn_fp8": True}))
    for i in range(B):              # default plans (chosen by a launch's tile count: split-K pairs, lx_gemm4_kernel): equal within rounding
        assert relerr(fwd(slice(i, i + 1), {"attn_fp8": True})[0].cpu(), v8[i].cpu()) < 5e-3
    assert torch.equal(ref, fwd(slice(None), {}))                        # switching back restores the bf16 path bit for bit
    eng.pair_plan = False           # the batch-size-invariant plans: a shard reproduces the batch bit for bit
    v8 = fwd(slice(None), {"attn_fp8": True})
    for i in range(B):
        assert torch.equal(fwd(slice(i, i + 1), {"attn_fp8": True})[0], v8[i])


@pytest.mark.parametrize("variant", ["pow2", "generic_scale"])
def test_fp8_attention_kernel_variants_agree(variant):
    """The pipelined fp8 kernel's two score paths (MX-block-scaled MFMAs with exp2 straight on the result when scale x log2 e x
    descale is a power of two -- the default, arranged by ops.FP8_Q_SCALE -- and the generic one-fma-per-score path taken for any
    other softmax scale), against fp32 SDPA."""
    import os, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["LX_T_SCALE"] = "0.1" if variant == "generic_scale" else ""
    code = r'''
import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from loongx_amd import ops
from tests.test_kernels_gpu import _qkv_buffer, _segments
from tests.helpers import relerr
B, H, lens = 2, 2, (64, 200, 320)
D = H * 128
buf = _qkv_buffer(B, lens, H, seed=9)
row0, vt0, vt_len = _segments(B, lens)
M = buf.shape[0]
Q8 = torch.zeros(M, D, dtype=torch.uint8, device="cuda"); K8 = torch.zeros_like(Q8)
VT8 = torch.zeros(B, H, 128, vt_len, dtype=torch.uint8, device="cuda")
ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, [(row0[s], lens[s], vt0[s], None, None, None, None) for s in range(3)], B, H, Q8, K8, VT8)
O = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
sc = float(os.environ["LX_T_SCALE"]) if os.environ.get("LX_T_SCALE") else None
ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, scale=sc)
x = buf.float().cpu()
def gather(col):
    out = torch.zeros(B, sum(lens), H, 128); p = 0
    for s, Ls in enumerate(lens):
        out[:, p:p + Ls] = x[row0[s]: row0[s] + B * Ls, col: col + D].view(B, Ls, H, 128); p += Ls
    return out.permute(0, 2, 1, 3)
q, k, v = gather(2 * D), gather(0), gather(D)
ref = torch.softmax((q @ k.transpose(-1, -2)) * (sc if sc else 1 / math.sqrt(128)), -1) @ v
got = torch.zeros_like(ref); p = 0
for s, Ls in enumerate(lens):
    got[:, :, p:p + Ls] = O.float().cpu()[row0[s]: row0[s] + B * Ls].view(B, Ls, H, 128).permute(0, 2, 1, 3); p += Ls
e = relerr(got, ref)
print("RELERR", e)
assert e < 7e-2, e
'''
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "RELERR" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


# ---- LX_EPI_QKV with e4m3 outputs: the projection epilogue writes the fp8 attention kernel's operand images itself ------------------
def _dq(img8, scale):
    return img8.view(torch.float8_e4m3fn).float() / scale


@pytest.mark.parametrize("bm", [256, 128])
def test_gemm_qkv_epilogue_e4m3_images(ops, monkeypatch, bm):
    """The byte images of the fused epilogue (q / k after RMSNorm + RoPE, V^T in the f8f6f4 operand order) against (a) an fp64
    restatement of block.py:43-99 rounded once to e4m3 and (b) the images lx_qkv_prep_fp8_segs makes from the bf16 projection
    (two roundings): the fused values are at least as close to (a) as the two-pass ones, the V^T bytes sit where the two-pass
    kernel puts them, and the bf16 outputs are NOT written."""
    from oracle.flux_modules import apply_rotary_emb, rope_tables
    from loongx_amd import _lib
    monkeypatch.setenv("LX_GEMM_BM", str(bm))          # (the 8-wave kernels own this epilogue; lx_gemm4_kernel's copy of it went in round 5)
    _lib.lib.lx_gemm_reload_env()
    B, H, K = 2, 2, 192
    D = H * 128
    N = 3 * D + 512
    lens = [96, 160]
    row0, vt0, vt_ld = _segments(B, lens)
    M = B * sum(lens)
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    Ws = [rnd(N, K, seed=2 + i, scale=K ** -0.5, dtype=torch.bfloat16) for i in range(2)]
    bias = [rnd(N, seed=4 + i, scale=0.3) for i in range(2)]
    wq = [1 + 0.1 * rnd(128, seed=6 + i) for i in range(2)]
    wk = [1 + 0.1 * rnd(128, seed=8 + i) for i in range(2)]
    tabs, cs_dev = [], []
    for i, Ls in enumerate(lens):
        ids = torch.zeros(Ls, 3)
        ids[:, 1] = torch.arange(Ls) // 8 + i
        ids[:, 2] = torch.arange(Ls) % 8 - 3 * i
        cos, sin = rope_tables(ids)
        tabs.append((cos, sin))
        cs = torch.empty(Ls, 128)
        cs[:, 0::2], cs[:, 1::2] = cos[:, 0::2], sin[:, 0::2]
        cs_dev.append(cs.to(DEV))
    u8 = torch.uint8

    def run(fused):
        C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        Q8, K8 = torch.zeros(M, D, dtype=u8, device=DEV), torch.zeros(M, D, dtype=u8, device=DEV)
        VT8 = torch.zeros(B, H, 128, vt_ld, dtype=u8, device=DEV)
        probs = []
        for i, Ls in enumerate(lens):
            rows = slice(row0[i], row0[i] + B * Ls)
            kw = {}
            if fused:
                kw["qkv"] = dict(norm_q=wq[i], norm_k=wk[i], rope=cs_dev[i], vt=VT8, vt_pos0=vt0[i], d=D, q8=Q8[rows], k8=K8[rows])
            probs.append(ops.gemm_desc(A[rows], Ws[i], C[rows], bias=bias[i], epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, rows_per_batch=Ls,
                                       gelu_col_start=3 * D, **kw))
        ops.gemm(probs)
        if not fused:
            segs = [(row0[i], Ls, vt0[i], wq[i], wk[i], tabs[i][0].to(DEV), tabs[i][1].to(DEV)) for i, Ls in enumerate(lens)]
            ops.qkv_prep_fp8_segs(C, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
        torch.cuda.synchronize()
        return C.float().cpu(), Q8.cpu(), K8.cpu(), VT8.cpu()
    Cf, Qf, Kf, Vf = run(True)
    Cu, Qu, Ku, Vu = run(False)
    assert torch.equal(Cf[:, 3 * D:], Cu[:, 3 * D:])                    # the GELU columns behind the projections: untouched by the flag
    assert bool((Cf[:, :3 * D] == 7.0).all())                          # no bf16 k / v / q is written in this form
    e4 = lambda x: x.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    for i, Ls in enumerate(lens):
        rows = slice(row0[i], row0[i] + B * Ls)
        y = A[rows].double().cpu() @ Ws[i].double().cpu().T + bias[i].double().cpu()
        cos, sin = (t.double() for t in tabs[i])

        def ref(x, w):
            x = x.view(B, Ls, -1, 128).permute(0, 2, 1, 3)
            x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double().cpu()
            return apply_rotary_emb(x, (cos, sin)).permute(0, 2, 1, 3).reshape(B * Ls, -1)
        for img_f, img_u, want, sc in ((Qf, Qu, ref(y[:, 2 * D:3 * D], wq[i]), ops.FP8_Q_SCALE), (Kf, Ku, ref(y[:, :D], wk[i]), ops.FP8_K_SCALE)):
            ef, eu = relerr(_dq(img_f[rows], sc), want), relerr(_dq(img_u[rows], sc), want)
            floor = relerr(e4(want.float() * sc) / sc, want)            # one e4m3 rounding of the exact value
            assert ef < 1.03 * floor + 1e-3 and ef <= eu * 1.02, (ef, eu, floor)
        # V^T bytes: same positions as the two-pass image; values = e4m3 of the fp32 accumulator (vs of its bf16 rounding)
        v = y[:, D:2 * D].view(B, Ls, H, 128).float()
        got, two = _dq(Vf, ops.FP8_V_SCALE), _dq(Vu, ops.FP8_V_SCALE)
        j = torch.arange(64)
        g_, p_ = j >> 5, j & 31
        key = (p_ >> 4) * 32 + 8 * ((p_ & 15) >> 2) + 4 * g_ + (p_ & 3)
        for t0 in range(0, Ls, 64):
            valid = key + t0 < Ls
            want_t = e4(v[:, (key[valid] + t0)]).permute(0, 2, 3, 1)     # [B, H, d, byte]
            sl = (slice(None), slice(None), slice(None), vt0[i] + t0 + j[valid])
            assert relerr(got[sl], want_t) < 2e-3 + 1.05 * relerr(two[sl], want_t), t0
            assert relerr(got[sl], two[sl]) < 4e-2


@pytest.mark.parametrize("mc", [{}, {"latent_lora": True}])
def test_engine_fp8_attention_with_fused_projection_epilogue(mc):
    """model_config attn_fp8 with bf16 GEMMs: the engine takes the e4m3 projection epilogue (no lx_qkv_prep_fp8_segs launch);
    against the fp32 oracle and against the same engine with qkv_epilogue = False (the two-pass path)."""
    from oracle import flux_modules as fm
    from oracle import flux_ref as fr
    from tests.helpers import tiny_transformer
    from tests.test_engine_gpu import _engine
    tr = tiny_transformer(seed=5)
    g = torch.Generator().manual_seed(7)
    B, T, hw = 2, 32, 8
    N = hw * hw
    kw = dict(hidden_states=torch.randn(B, N, 64, generator=g), encoder_hidden_states=torch.randn(B, T, 64, generator=g) * 0.5,
              pooled_projections=torch.randn(B, 32, generator=g), timestep=torch.tensor([0.8, 0.3]),
              img_ids=fm.prepare_latent_image_ids(hw, hw), txt_ids=torch.zeros(T, 3), guidance=torch.full((B,), 3.5))
    cond = torch.randn(B, N, 64, generator=g)
    cids = fm.prepare_latent_image_ids(hw, hw)
    cids[:, 2] -= hw
    with torch.no_grad():
        want = fr.tranformer_forward(tr, cond, cids, None, mc, **kw)[0]
    d = "cuda"
    mc8 = dict(mc, attn_fp8=True)
    outs = {}
    for fused in ("1", "0"):
        eng = _engine(tr)
        eng.qkv_epilogue = fused == "1"
        eng.set_conditioning(kw["encoder_hidden_states"].to(d), kw["pooled_projections"].to(d), kw["guidance"].to(d), kw["txt_ids"].to(d),
                             kw["img_ids"].to(d), cond.to(d), cids.to(d), c_t=0.0, model_config=mc8)
        assert eng._qkv_epilogue() == (fused == "1")
        a = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()
        b = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()      # graph replay
        assert torch.equal(a, b)
        outs[fused] = a
    e1, e0 = relerr(outs["1"], want), relerr(outs["0"], want)
    assert e1 < TOL_FP8 and e0 < TOL_FP8 and e1 < 1.25 * e0 + 2e-3, (e1, e0)
    assert relerr(outs["1"], outs["0"]) < TOL_FP8
