"""The fp16 operand mode (include/lx.h LX_OPERANDS_F16, LX_ATTN_O_F16; model_config["operands"] = "fp16" / dtype=torch.float16) on a
real MI355X: the producers of the fp16 operand images (saturation, the overflow word, nearest-even rounding), fp16 subnormal weights on
the matrix pipe, the projection epilogue with fp16 operands and bf16 q / k / V^T, the attention kernels' fp16 output, and the engine
against the fp32 oracle -- where the mode has to land 6-8x closer than the bf16 mode does (tools/bf16_ablation.py).
The GEMM plan / epilogue matrix itself runs in both formats in tests/test_kernels_gpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import relerr, tiny_transformer  # noqa: E402
from tests.test_kernels_gpu import DEV, _qkv_buffer, _segments, _attn_reference, ops, rnd, _ws  # noqa: E402,F401

F16_MAX = 65504.0


def test_ln_modulate_f16_rounds_to_nearest_and_reports_saturation(ops):
    """lx_ln_modulate_f16_segs: the same rows as the bf16 kernel to fp16 precision (bit-equal to torch's fp32 -> fp16 of the kernel's own
    fp32 arithmetic is not observable; against an fp64 restatement the error is the 11-bit rounding), and a row whose modulated value
    leaves fp16's range is clipped to +-65504 and counted."""
    B, Lr, D = 3, 37, 3072
    X = rnd(B * Lr, D, seed=1, scale=2.0) + 0.5
    sh, sc = rnd(B, D, seed=2), rnd(B, D, seed=3, scale=0.3)
    Yh = torch.empty(B * Lr, D, dtype=torch.float16, device=DEV)
    Yb = torch.empty(B * Lr, D, dtype=torch.bfloat16, device=DEV)
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.ln_modulate(X, sh, sc, Yh, rows_per_batch=Lr, f16_ovf=ovf)
    ops.ln_modulate(X, sh, sc, Yb, rows_per_batch=Lr)
    ref = (torch.nn.functional.layer_norm(X.double(), (D,), eps=1e-6).view(B, Lr, D) * (1 + sc.double()[:, None]) + sh.double()[:, None]).reshape(B * Lr, D)
    eh, eb = relerr(Yh.cpu(), ref.cpu()), relerr(Yb.cpu(), ref.cpu())
    assert eh < 2.5e-4 and eb > 5 * eh, (eh, eb)            # 2^-12 / sqrt(3) = 1.4e-4 against 2^-9 / sqrt(3) = 1.1e-3
    assert int(ovf) == 0
    # one modulation scale of 5e4 on batch 1: |LN(x)| up to ~4 -> 2e5 > 65504 on a handful of rows of that batch only
    sc2 = sc.clone()
    sc2[1, 7] = 5.0e4
    ops.ln_modulate(X, sh, sc2, Yh, rows_per_batch=Lr, f16_ovf=ovf)
    col = Yh[Lr:2 * Lr, 7].float()
    want = (ref[Lr:2 * Lr, 7] - sh[1, 7].double()) / (1 + sc[1, 7].double()) * (1 + 5.0e4) + sh[1, 7].double()
    clipped = want.abs() > F16_MAX
    assert bool(clipped.any()) and bool((~clipped).any())
    assert torch.equal(col[clipped].cpu(), torch.sign(want[clipped]).float().cpu() * F16_MAX)
    assert bool(torch.isfinite(Yh.float()).all())
    assert int(ovf) == int(clipped.sum())                   # one wave per row: the counter is the number of rows that clipped
    Yg = torch.empty(5, 1024, dtype=torch.float16, device=DEV)       # the generic-D kernel takes the same path
    Xg = rnd(5, 1024, seed=4)
    ops.ln_modulate(Xg, rnd(1, 1024, seed=5), rnd(1, 1024, seed=6, scale=0.2), Yg, rows_per_batch=5, f16_ovf=ovf)
    assert bool(torch.isfinite(Yg.float()).all())


def test_convert_and_lora_down_f16(ops):
    v32 = rnd(1000, seed=3, scale=300.0)
    v32[:4] = torch.tensor([7.0e4, -1.0e6, 65504.0, -65520.0])
    d = torch.empty(1000, dtype=torch.float16, device=DEV)
    ops.convert(d, v32)
    assert torch.equal(d, v32.clamp(-F16_MAX, F16_MAX).to(torch.float16))
    M, K, r = 300, 3072, 12
    A = rnd(M, K, seed=1, dtype=torch.bfloat16).to(torch.float16)
    Ad = rnd(r, K, seed=3, scale=K ** -0.5, dtype=torch.bfloat16).to(torch.float16)
    slabs = torch.full((4, M, 16), float("nan"), device=DEV)
    ops.lora_down(A, Ad, slabs[0, :, :r], n_split=4, split_stride=slabs.stride(0))
    assert relerr(slabs[:, :, :r].double().sum(0).cpu(), (A.double() @ Ad.double().T).cpu()) < 2e-6
    with pytest.raises(TypeError):
        ops.lora_down(A, Ad.to(torch.bfloat16), slabs[0, :, :r])      # the two operands share one format


@pytest.mark.parametrize("plan", ["8wave", "g4"])
def test_gemm_f16_store_saturates_and_counts(ops, monkeypatch, plan):
    """A 16-bit store of an fp16-operand launch: values beyond +-65504 are clipped (never inf), the counter says how many waves did, and
    every other element is the nearest-even fp16 of the fp32 result."""
    monkeypatch.setenv("LX_GEMM4", "2" if plan == "g4" else "0")
    ops.lib.lx_gemm_reload_env()
    M, N, K = 512, 512, 128
    A = rnd(M, K, seed=1, dtype=torch.bfloat16).to(torch.float16)
    W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16).to(torch.float16)
    bias = rnd(N, seed=3)
    bias[5], bias[300] = 1.0e5, -3.0e5                     # two whole output columns out of range
    C = torch.empty(M, N, dtype=torch.float16, device=DEV)
    C32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, C, bias=bias, f16=True, f16_ovf=ovf)])
    ops.gemm([ops.gemm_desc(A, W, C32, bias=bias, epilogue=ops.LX_EPI_STORE_F32, f16=True)])
    assert bool(torch.isfinite(C.float()).all())
    assert bool((C[:, 5] == F16_MAX).all()) and bool((C[:, 300] == -F16_MAX).all())
    assert torch.equal(C, C32.clamp(-F16_MAX, F16_MAX).to(torch.float16))      # same accumulation, one nearest-even rounding
    n = int(ovf)
    assert 0 < n <= 2 * (M // 16) * 4, n                   # waves that saw a clipped value (their count depends on the kernel's wave tiling)
    ovf.zero_()
    bias[5], bias[300] = 0.0, 0.0
    ops.gemm([ops.gemm_desc(A, W, C, bias=bias, f16=True, f16_ovf=ovf)])
    assert int(ovf) == 0


def test_gemm_f16_subnormal_weights_are_honoured(ops):
    """bf16 weights below 2^-14 convert to fp16 SUBNORMALS (exactly, down to 2^-24 granularity): the matrix pipe has to multiply them, not
    flush them -- 0.24 % of N(0, 0.02^2) weights live there. Products against fp64 at fp32-accumulation accuracy."""
    M, N, K = 256, 256, 512
    A = rnd(M, K, seed=1, dtype=torch.bfloat16).to(torch.float16)
    Wb = (rnd(N, K, seed=2, dtype=torch.bfloat16).float() * 2.0 ** -18).to(torch.bfloat16)     # |w| ~ 4e-6 ... 2e-5: all subnormal in fp16
    W = Wb.to(torch.float16)
    assert float(W.float().abs().max()) < 2.0 ** -14 and float(W.float().abs().max()) > 0
    C = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, C, epilogue=ops.LX_EPI_STORE_F32, f16=True)])
    ref = A.double() @ W.double().T
    assert relerr(C.cpu(), ref.cpu()) < 2e-6
    assert relerr(W.double().cpu(), Wb.double().cpu()) < 2e-2      # (what the conversion itself lost at this extreme scale: 2^-24 granularity)


@pytest.mark.parametrize("bm", [256, 128, None])
def test_gemm_qkv_epilogue_with_f16_operands_keeps_bf16_attention_operands(ops, monkeypatch, bm):
    """LX_EPI_QKV on an fp16-operand launch (the single block's fused [k | v | q | mlp] projection): k and q (RMSNorm + RoPE) and the V^T
    image come out in bf16 exactly as from the bf16-operand launch's epilogue fed the same accumulators, the GELU columns behind them
    in fp16."""
    from oracle.flux_modules import rope_tables
    if bm is not None:
        monkeypatch.setenv("LX_GEMM_BM", str(bm))
        ops.lib.lx_gemm_reload_env()
    B, H, K, Ls = 2, 2, 192, 96
    D = H * 128
    N = 3 * D + 512
    M = B * Ls
    Ab = rnd(M, K, seed=1, dtype=torch.bfloat16)
    Wb = rnd(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    bias = rnd(N, seed=4, scale=0.3)
    wq, wk = 1 + 0.1 * rnd(128, seed=6), 1 + 0.1 * rnd(128, seed=8)
    ids = torch.zeros(Ls, 3)
    ids[:, 1], ids[:, 2] = torch.arange(Ls) // 8, torch.arange(Ls) % 8
    cos, sin = rope_tables(ids)
    cs = torch.empty(Ls, 128)
    cs[:, 0::2], cs[:, 1::2] = cos[:, 0::2], sin[:, 0::2]
    cs = cs.to(DEV)

    def run(f16):
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        VT = torch.zeros(B, H, 128, 128, dtype=torch.bfloat16, device=DEV)
        A, W = (Ab.to(torch.float16), Wb.to(torch.float16)) if f16 else (Ab, Wb)
        kw = dict(f16=True) if f16 else {}
        ops.gemm([ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, rows_per_batch=Ls, gelu_col_start=3 * D,
                                qkv=dict(norm_q=wq, norm_k=wk, rope=cs, vt=VT, vt_pos0=0, d=D), **kw)])
        torch.cuda.synchronize()
        return C, VT
    Cb, VTb = run(False)
    Ch, VTh = run(True)
    # the operands are bf16-representable in both launches, so the accumulators agree bit for bit and so do the bf16 outputs
    assert torch.equal(Cb[:, :D], Ch[:, :D]) and torch.equal(Cb[:, 2 * D:3 * D], Ch[:, 2 * D:3 * D]) and torch.equal(VTb, VTh)
    y = Ab.double() @ Wb.double().T + bias.double()
    g = torch.nn.functional.gelu(y[:, 3 * D:], approximate="tanh")
    eh = relerr(Ch.view(torch.float16)[:, 3 * D:].cpu(), g.cpu())
    eb = relerr(Cb[:, 3 * D:].cpu(), g.cpu())
    assert eh < 3e-4 and eb > 4 * eh, (eh, eb)


@pytest.mark.parametrize("kernel", ["8wave", "4wave", "fp8"])
def test_attention_output_as_fp16(ops, kernel):
    """LX_ATTN_O_F16: the same attention with O written as fp16 -- bf16 O and fp16 O are two roundings of the same fp32 rows (the fp16
    one 8x closer to them), on the 8-wave kernel, the one-wave-per-SIMD kernel and the e4m3 kernel."""
    lens = (64, 128, 200)
    B, H = 2, 3
    D = H * 128
    row0, vt0, vt_len = _segments(B, lens)
    if kernel == "fp8":
        buf = _qkv_buffer(B, lens, H, seed=5)
        segs = [(row0[s], Ls, vt0[s], None, None, None, None) for s, Ls in enumerate(lens)]
        Q8 = torch.zeros(buf.shape[0], D, dtype=torch.uint8, device=DEV); K8 = torch.zeros_like(Q8)
        VT8 = torch.zeros(B, H, 128, vt_len, dtype=torch.uint8, device=DEV)
        ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
        outs = {}
        for f16 in (False, True):
            O = torch.zeros(buf.shape[0], D, dtype=torch.float16 if f16 else torch.bfloat16, device=DEV)
            ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0,
                             flags=ops.ATTN_O_F16 if f16 else 0, f16_ovf=ovf if f16 else None)
            outs[f16] = O.float().cpu()
            assert int(ovf) == 0
        assert relerr(outs[True], outs[False]) < 3e-3        # two roundings of the same rows
        return
    if kernel == "4wave":
        # the planner's own choice (library switches are read once per process): a bounded-score launch of >= 2 rounds of workgroups with
        # short items goes to lx_attn4_kernel. Same rows in both formats; the fp32-reference comparison is the 8-wave arm's.
        B, H = 8, 24
        D = H * 128
        row0, vt0, vt_len = _segments(B, lens)
        buf = _qkv_buffer(B, lens, H, seed=5)
        one = torch.ones(128, device=DEV)
        VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=DEV)
        ops.qkv_prep_segs(buf, 2 * D, 0, D, [(row0[s], Ls, vt0[s], one * ops.Q_LOG2_FACTOR, one, None, None) for s, Ls in enumerate(lens)], B, H, VT)
        outs = {}
        for f16 in (False, True):
            O = torch.zeros(buf.shape[0], D, dtype=torch.float16 if f16 else torch.bfloat16, device=DEV)
            ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
            ops.attn_fwd(buf, buf, VT, O, q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0,
                         flags=ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED | (ops.ATTN_O_F16 if f16 else 0), f16_ovf=ovf if f16 else None)
            assert ops.lib.lx_attn_last_kernel() == 2              # LX_ATTN_KERNEL_4WAVE
            outs[f16] = O.float().cpu()
            assert int(ovf) == 0
        e = relerr(outs[True], outs[False])
        assert 1e-4 < e < 3e-3, e                                   # the bf16 rounding of the same fp32 rows
        # (re-rounding the fp16 rows to bf16 does NOT reproduce the bf16 store: one value in eight is a tie at fp16 precision)
        return
    buf = _qkv_buffer(B, lens, H, seed=5)
    VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=DEV)
    for s, Ls in enumerate(lens):
        ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=row0[s], n_rows=B * Ls, rows_per_batch=Ls, H=H, wq=None, wk=None,
                     cos=None, sin=None, VT=VT, vt_pos0=vt0[s])
    ref, edges = _attn_reference(buf, B, H, lens, [[0.0] * 3] * 3, 2 * D, 0, D)
    outs = {}
    for f16 in (False, True):
        O = torch.zeros(buf.shape[0], D, dtype=torch.float16 if f16 else torch.bfloat16, device=DEV)
        ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.attn_fwd(buf, buf, VT, O, q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0,
                     flags=ops.ATTN_O_F16 if f16 else 0, f16_ovf=ovf if f16 else None)
        outs[f16] = O.float().cpu()
        assert int(ovf) == 0
    errs = {}
    for f16 in (False, True):
        e = []
        for s, Ls in enumerate(lens):
            o = outs[f16][row0[s]: row0[s] + B * Ls].view(B, Ls, H, 128)
            e.append(relerr(o, ref[:, edges[s]:edges[s + 1]]))
        errs[f16] = max(e)
    assert errs[False] < 6e-3 and errs[True] < errs[False], errs        # (both carry the bf16 rounding of P; O's own rounding shrinks)
    assert relerr(outs[True], outs[False]) < 3e-3


def _tiny_pair(seed=0):
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    tr = tiny_transformer(seed)
    with torch.no_grad():                 # bf16-representable base weights, as FLUX.1 checkpoints hold them: the comparison then measures
        for name, p_ in tr.named_parameters():      # the arithmetic of the two operand formats, not the (shared) bf16 rounding of the weights
            if p_.dim() >= 2 and ".lora_" not in name:
                p_.copy_(p_.to(torch.bfloat16).float())
    cfg = FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, in_channels=64, joint_attention_dim=64,
                     pooled_projection_dim=32, guidance_embeds=True)
    return tr, LxFluxTransformer.from_state_dict(tr.state_dict(), cfg, "cuda:0")


@pytest.mark.parametrize("mc", [{}, {"latent_lora": True}, {"independent_condition": True}, {"attn_fp8": True}])
def test_tiny_forward_f16_operands_against_the_oracle(ops, mc):
    """The 2 + 2-block three-stream forward with LoRA on the condition stream, fp16 operand images, against the fp32 oracle: at least 4x
    closer than the bf16 mode on the same inputs (attn_fp8: the e4m3 attention dominates both), no saturation."""
    from oracle import flux_modules as fm
    from oracle import flux_ref as fr
    from loongx_amd.flux.transformer import tranformer_forward
    tr, lx = _tiny_pair()
    g = torch.Generator().manual_seed(1)
    B, T, hw = 2, 32, 8
    kw = dict(hidden_states=torch.randn(B, hw * hw, 64, generator=g), encoder_hidden_states=torch.randn(B, T, 64, generator=g) * 0.5,
              pooled_projections=torch.randn(B, 32, generator=g), timestep=torch.tensor([0.7, 0.35]),
              img_ids=fm.prepare_latent_image_ids(hw, hw), txt_ids=torch.zeros(T, 3), guidance=torch.full((B,), 3.5))
    cond = torch.randn(B, hw * hw, 64, generator=g)
    cids = fm.prepare_latent_image_ids(hw, hw)
    cids[:, 2] -= hw
    omc = {k: v for k, v in mc.items() if k != "attn_fp8"}
    with torch.no_grad():
        want = fr.tranformer_forward(tr, cond, cids, None, omc, **kw)[0]
    errs = {}
    for fmt in ("bf16", "fp16"):
        lx.invalidate_conditioning()
        got = tranformer_forward(lx, cond.cuda(), cids.cuda(), None, dict(mc, operands=fmt), return_dict=False, **{k: v.cuda() for k, v in kw.items()})[0]
        errs[fmt] = relerr(got.cpu(), want)
        # the second forward of the same conditioning replays the captured graph (and, with independent_condition, the cached condition stream)
        got2 = tranformer_forward(lx, cond.cuda(), cids.cuda(), None, dict(mc, operands=fmt), return_dict=False, **{k: v.cuda() for k, v in kw.items()})[0]
        assert relerr(got2.cpu(), want) < 1.05 * errs[fmt] + 1e-6
    assert lx.engine.f16 and lx.engine.f16_overflow_count() == 0
    if mc.get("attn_fp8"):
        assert errs["fp16"] < errs["bf16"] and errs["fp16"] < 7e-2, errs
    else:
        assert errs["fp16"] < 1.2e-3 and errs["fp16"] < errs["bf16"] / 4, errs
    assert lx.engine.w16_inexact_share < 1e-3


@pytest.mark.parametrize("case", ["epilogue_off", "ragged", "ragged_attn_fp8"])
def test_f16_operands_without_the_fused_projection_epilogue(ops, case):
    """The fp16 operand mode where LX_EPI_QKV is not taken (round-5 advisor finding: the separate pass used to read the fp16 k | v | q
    store as bf16): qkv_epilogue switched off at fused-capable shapes, and stream lengths that are no multiple of 32 (T = 24 text tokens,
    a 6 x 5 latent grid: 30 image + 30 condition tokens, as 720 x 480 gives 1350). lx_qkv_prep_f16in_segs / lx_qkv_prep_fp8_f16in_segs
    read the projection as fp16; the result has to land where the fused form lands against the fp32 oracle."""
    from oracle import flux_modules as fm
    from oracle import flux_ref as fr
    from loongx_amd.flux.transformer import tranformer_forward
    tr, lx = _tiny_pair(seed=5)
    g = torch.Generator().manual_seed(4)
    ragged = case.startswith("ragged")
    B, T, gh, gw = 2, (24 if ragged else 32), (6 if ragged else 8), (5 if ragged else 8)
    N = gh * gw
    mc = {"attn_fp8": True} if case.endswith("attn_fp8") else {}
    kw = dict(hidden_states=torch.randn(B, N, 64, generator=g), encoder_hidden_states=torch.randn(B, T, 64, generator=g) * 0.5,
              pooled_projections=torch.randn(B, 32, generator=g), timestep=torch.tensor([0.6, 0.2]),
              img_ids=fm.prepare_latent_image_ids(gh, gw), txt_ids=torch.zeros(T, 3), guidance=torch.full((B,), 3.5))
    cond = torch.randn(B, N, 64, generator=g)
    cids = fm.prepare_latent_image_ids(gh, gw)
    cids[:, 2] -= gw
    with torch.no_grad():
        want = fr.tranformer_forward(tr, cond, cids, None, {}, **kw)[0]
    lx.engine.qkv_epilogue = case != "epilogue_off"
    errs = {}
    try:
        for fmt in ("bf16", "fp16"):
            lx.invalidate_conditioning()
            got = tranformer_forward(lx, cond.cuda(), cids.cuda(), None, dict(mc, operands=fmt), return_dict=False, **{k: v.cuda() for k, v in kw.items()})[0]
            assert not lx.engine._qkv_epilogue()                       # the separate pass is what ran
            assert torch.isfinite(got).all()
            errs[fmt] = relerr(got.cpu(), want)
        assert lx.engine.f16 and lx.engine.f16_overflow_count() == 0
    finally:
        lx.engine.qkv_epilogue = True
    if mc:
        assert errs["fp16"] < errs["bf16"] and errs["fp16"] < 7e-2, errs
    else:
        # the unfused form rounds k / q twice (fp16 store, then bf16 after RMSNorm + RoPE): the bound of the fused form, a little wider
        assert errs["fp16"] < 1.5e-3 and errs["fp16"] < errs["bf16"] / 3, errs


def test_f16_default_from_dtype_and_block_level_mirrors(ops):
    """operands_default = "fp16" (what dtype=torch.float16 selects) runs the mode without a model_config entry, a call can still ask for
    bf16, and the engine's block-level accessors read the fp16 images as such."""
    from oracle import flux_modules as fm
    from loongx_amd.flux.transformer import tranformer_forward
    tr, lx = _tiny_pair(seed=3)
    lx.engine.operands_default = "fp16"
    g = torch.Generator().manual_seed(2)
    B, T, hw = 1, 32, 8
    kw = dict(hidden_states=torch.randn(B, hw * hw, 64, generator=g).cuda(), encoder_hidden_states=(torch.randn(B, T, 64, generator=g) * 0.5).cuda(),
              pooled_projections=torch.randn(B, 32, generator=g).cuda(), timestep=torch.tensor([0.5]).cuda(),
              img_ids=fm.prepare_latent_image_ids(hw, hw).cuda(), txt_ids=torch.zeros(T, 3).cuda(), guidance=torch.full((B,), 3.5).cuda())
    a = tranformer_forward(lx, None, None, None, {}, return_dict=False, **kw)[0].clone()
    assert lx.engine.f16
    lx.invalidate_conditioning()
    b = tranformer_forward(lx, None, None, None, {"operands": "bf16"}, return_dict=False, **kw)[0].clone()
    assert not lx.engine.f16
    e = relerr(a.cpu(), b.cpu())
    assert 1e-5 < e < 2e-2, e                                # two operand formats of the same forward
    with pytest.raises(ValueError):
        lx.invalidate_conditioning()
        tranformer_forward(lx, None, None, None, {"operands": "fp8"}, return_dict=False, **kw)


# ---- the PRODUCT's handling of a saturated fp16 operand (generate() / model_config["f16_overflow"]) ---------------------------------------
def _planted_model(bias_value=None, weight_value=None):
    """The tiny three-stream model behind generate(), with one value planted through the WEIGHTS that leaves fp16's range on the way:
    an ff.net[0] bias of 1e5 (the MLP hidden's 16-bit store saturates at run time), or a q/k/v weight of 1e5 (the weight image itself)."""
    from oracle import cs3 as ocs3
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    from loongx_amd.train.model import OminiModel
    from oracle import flux_modules as fm
    tr = fm.FluxTransformer2DModel(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=4096,
                                   pooled_dim=768, guidance_embeds=True, lora=True)
    fm.init_synthetic_(tr, seed=4, std=0.03, bias_std=0.02, norm_jitter=0.1)
    with torch.no_grad():
        if bias_value is not None:
            tr.transformer_blocks[0].ff.net[0].proj.bias[5] = bias_value
        if weight_value is not None:
            lin = tr.transformer_blocks[1].attn.to_q
            getattr(lin, "base_layer", lin).weight[3, 7] = weight_value
    cfg = FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, in_channels=64, joint_attention_dim=4096,
                     pooled_projection_dim=768, guidance_embeds=True)
    torch.manual_seed(0)
    cs3 = ocs3.CS3DGF(seed=0).eval()
    lxtr = LxFluxTransformer.from_state_dict(tr.eval().state_dict(), cfg, "cuda")
    return OminiModel.from_pipe(LxFluxPipeline(lxtr), cs3.state_dict(), {}, "cuda")


def _gen(model, mc, seed=3):
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    g = torch.Generator().manual_seed(seed)
    hw = 8
    lat, cond = torch.randn(1, hw * hw, 64, generator=g), torch.randn(1, hw * hw, 64, generator=g)
    pe, pooled = torch.randn(1, 512, 4096, generator=g) * 0.1, torch.randn(1, 768, generator=g)
    c = Condition("subject", latents=cond.cuda(), latent_hw=(hw, hw), position_delta=[0, -hw])
    return generate(model, model.flux_pipe, conditions=[c], height=hw * 16, width=hw * 16, num_inference_steps=4, latents=lat.cuda(),
                    prompt_embeds=pe.cuda(), pooled_prompt_embeds=pooled.cuda(), output_type="latent", model_config=mc, default_lora=True,
                    use_brain_condition=False)


def test_generate_applies_the_f16_overflow_policy_to_a_saturated_activation(ops):
    """Round-5 verdict: the kernels saturate and count, but the PRODUCT never read the counter. A bias of 1e5 on ff.net[0] of the first
    double block drives one MLP-hidden channel past 65504: under "raise" generate() (latent output: asynchronous read) surfaces it at the
    next image at the latest and the synchronous poll at once; "warn" keeps the image and says so; "fallback" recomputes the image with
    bf16 operands -- bit-equal to asking for bf16 in the first place. A clean model raises nothing under any policy."""
    import warnings as _w
    from loongx_amd.flux.generate import F16OverflowError
    model = _planted_model(bias_value=1.0e5)
    eng = model.transformer.engine
    want_bf16 = _gen(model, {"operands": "bf16"}).images.clone()
    assert not eng.w16                                            # the bf16 call left no fp16 weight images behind
    # raise: the first image returns (its counter read is in flight), the synchronous poll -- what inference.py does before saving -- reports
    out = _gen(model, {"operands": "fp16", "f16_overflow": "raise"})
    assert out.lx_operands == "fp16" and eng.f16
    torch.cuda.synchronize()
    with pytest.raises(F16OverflowError):
        _gen(model, {"operands": "fp16", "f16_overflow": "raise"})         # the landed read of image 1 (and image 2's own events)
    assert eng.f16_overflow_poll(sync=True) > 0                             # image 2 ran: its events are there for the synchronous reader
    assert eng.f16_overflow_poll(sync=True) == 0                            # ... and were consumed
    # warn: synchronous path not needed -- the second call sees the first one's read
    with _w.catch_warnings(record=True) as rec:
        _w.simplefilter("always")
        _gen(model, {"operands": "fp16", "f16_overflow": "warn"})
        torch.cuda.synchronize()
        out = _gen(model, {"operands": "fp16", "f16_overflow": "warn"})
    assert any("saturation" in str(r.message) for r in rec) and out.lx_operands == "fp16"
    eng.f16_overflow_poll(sync=True)
    # fallback: decided per image, synchronously
    with pytest.warns(RuntimeWarning, match="recomputing this image with bf16 operands"):
        out = _gen(model, {"operands": "fp16", "f16_overflow": "fallback"})
    assert out.lx_operands.startswith("bf16") and torch.equal(out.images, want_bf16)
    with pytest.raises(ValueError):
        _gen(model, {"operands": "fp16", "f16_overflow": "ignore"})
    # inference.py's pre-save check
    import inference as inf
    _gen(model, {"operands": "fp16", "f16_overflow": "warn"})
    with pytest.raises(F16OverflowError):
        inf._f16_checked(eng, {"f16_overflow": "raise"})
    # a clean model: no event under any policy, the image is the fp16 one
    clean = _planted_model()
    with _w.catch_warnings():
        _w.simplefilter("error")
        for pol in ("raise", "warn", "fallback"):
            o = _gen(clean, {"operands": "fp16", "f16_overflow": pol})
            torch.cuda.synchronize()
            assert o.lx_operands == "fp16"
        assert clean.transformer.engine.f16_overflow_poll(sync=True) == 0


def test_a_weight_beyond_fp16_range_is_saturated_counted_and_reported(ops):
    """W.to(float16) turned |w| > 65504 into inf (then NaN) without a trace. The weight images saturate, `w16_clipped`
    counts, and generate() reports it under the same policy -- on the FIRST image (the count is known on the host)."""
    from loongx_amd.flux.generate import F16OverflowError
    model = _planted_model(weight_value=1.0e5)
    eng = model.transformer.engine
    with pytest.raises(F16OverflowError, match="clipped weights: 1"):
        _gen(model, {"operands": "fp16", "f16_overflow": "raise"})
    assert eng.w16_clipped == 1 and all(torch.isfinite(v.float()).all() for v in eng.w16.values())
    with pytest.warns(RuntimeWarning):
        out = _gen(model, {"operands": "fp16", "f16_overflow": "fallback"})
    assert out.lx_operands.startswith("bf16") and torch.isfinite(out.images).all()
