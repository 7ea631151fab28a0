"""fp8 (OCP e4m3) GEMM path -- BASELINE configs[4], opt-in `model_config["gemm_fp8"]`: lx_gemm_bf16 with LX_OPERANDS_FP8 and the
producers of its operand images (csrc/gemm.hip lx_gemm_fp8_kernel, csrc/fp8.hip).

Two kinds of statement:
  * the KERNELS are exact: given e4m3 operand bytes, the GEMM equals the fp32 product of the de-quantised operands (fp32
    accumulation order aside, 2e-5), the converters round to nearest-even like torch's float8_e4m3fn cast;
  * the MODE has the error of e4m3 itself (3 mantissa bits: ~3.6 % rms per element, ~5 % per GEMM output with both operands
    rounded) -- measured on the 4-block model against the fp32 reference goldens and printed; the assert bounds it.
"""
import json

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
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def deq(u8):
    return u8.view(torch.float8_e4m3fn).float()


def q8(x, scale):
    return (x * scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


def test_convert_and_ln_fp8_match_torch_cast(ops):
    x = rnd(300, 512, seed=1, scale=3.0)
    d = torch.zeros(300, 640, dtype=torch.uint8, device=DEV)
    ops.convert_fp8(x, d[:, :512], 16.0)
    assert torch.equal(d[:, :512], q8(x, 16.0))
    xb = x.to(torch.bfloat16)
    ops.convert_fp8(xb, d[:, :512], 2.0)
    assert torch.equal(d[:, :512], q8(xb.float(), 2.0))
    ops.convert_fp8(x * 1000, d[:, :512], 16.0)                       # saturates instead of producing NaN
    assert float(deq(d[:, :512]).abs().max()) == 448.0
    M, D = 200, 3072
    X = rnd(M, D, seed=2, scale=2.0)
    sh, sc = rnd(2, D, seed=3, scale=0.3), rnd(2, D, seed=4, scale=0.3)
    Y = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
    Y8 = torch.zeros(M, D, dtype=torch.uint8, device=DEV)
    ops.ln_modulate_fp8_segs(X, [(0, M, 100, sh, sc)], Y, Y8, D, 16.0)
    Yref = torch.zeros_like(Y)
    ops.ln_modulate_segs(X, [(0, M, 100, sh, sc)], Yref, D)
    assert relerr(Y.float(), Yref.float()) < 4e-3                        # (3-pass vs register-resident statistics)
    ref = torch.nn.functional.layer_norm(X, (D,), eps=1e-6) * (1 + sc.repeat_interleave(100, 0)) + sh.repeat_interleave(100, 0)
    assert relerr(deq(Y8) / 16.0, ref) < 4e-2                            # e4m3 rounding: 3.6 % rms
    assert (deq(Y8) / 16.0 - ref).abs().max() < 0.07 * ref.abs().max()


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 384), (2560, 768, 3072), (1000, 9216, 3072)])
def test_gemm_fp8_equals_dequantised_product(ops, M, N, K, monkeypatch):
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.02)
    bias = rnd(N, seed=3)
    A8 = q8(A, 16.0)
    W8, rs = ops.quantize_weight_fp8(W)
    assert relerr(deq(W8) * rs[:, None], W) < 4e-2
    cs = (rs / 16.0).contiguous()
    ref = (deq(A8).double() @ deq(W8).double().T * cs.double() + bias.double()).float()
    assert relerr(ref, A @ W.T + bias) < 8e-2                            # what e4m3 operands cost on this product
    for bm in ("256", "128"):
        monkeypatch.setenv("LX_GEMM_BM", bm)
        ops.lib.lx_gemm_reload_env()
        for tiled in (False, True):
            if tiled and N % 256:
                continue
            Wd = ops.tile_weight(W8) if tiled else W8
            C32 = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm([ops.gemm_desc(A8, Wd, C32, bias=bias, epilogue=ops.LX_EPI_STORE_F32, fp8=True, col_scale=cs)])
            assert relerr(C32, ref) < 2e-5, (bm, tiled)
    monkeypatch.delenv("LX_GEMM_BM")
    ops.lib.lx_gemm_reload_env()
    assert torch.equal(ops.untile_weight(ops.tile_weight(W8)), W8) if N % 256 == 0 else True
    # bf16 store; gated fp32 residual; e4m3 store with GELU
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm([ops.gemm_desc(A8, W8, Cb, bias=bias, fp8=True, col_scale=cs)])
    assert relerr(Cb.float(), ref) < 4e-3
    X0, gate = rnd(M, N, seed=5), rnd(1, N, seed=6)
    X = X0.clone()
    ops.gemm([ops.gemm_desc(A8, W8, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, fp8=True, col_scale=cs)])
    assert relerr(X, X0 + gate * ref) < 2e-5
    C8 = torch.zeros(M, N + 64, dtype=torch.uint8, device=DEV)
    ops.gemm([ops.gemm_desc(A8, W8, C8[:, :N], bias=bias, epilogue=ops.LX_EPI_STORE_FP8 | ops.LX_EPI_GELU, fp8=True, col_scale=cs, out_scale=16.0)])
    g = torch.nn.functional.gelu(ref, approximate="tanh")
    want8 = q8(g, 16.0)
    got, want = deq(C8[:, :N]), deq(want8)
    assert relerr(got, want) < 2e-2 and float((got != want).float().mean()) < 0.02     # identical up to values on a rounding boundary


def test_gemm_fp8_lora_and_lora_down_fp8(ops):
    M, N, K, r = 520, 512, 1024, 4
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.02)
    A8 = q8(A, 16.0)
    W8, rs = ops.quantize_weight_fp8(W)
    cs = (rs / 16.0).contiguous()
    Ad = rnd(r, K, seed=3, scale=0.1).to(torch.bfloat16)
    Bu = rnd(N, r, seed=4, scale=0.2)
    slabs = torch.full((4, M, 16), float("nan"), device=DEV)
    ops.lora_down_fp8(A8, 1.0 / 16.0, Ad, slabs[0, :, :r], n_split=4, split_stride=slabs.stride(0))
    t = (deq(A8) / 16.0) @ Ad.float().T
    assert relerr(slabs[:, :, :r].sum(0), t) < 1e-5
    C32 = torch.empty(M, N, device=DEV)
    ops.gemm([ops.gemm_desc(A8, W8, C32, epilogue=ops.LX_EPI_STORE_F32, fp8=True, col_scale=cs, lora_t=slabs[0, :, :r], lora_up=Bu, lora_nsplit=4,
                            lora_split_stride=slabs.stride(0))])
    ref = deq(A8) @ deq(W8).T * cs + t @ Bu.T
    assert relerr(C32, ref) < 1e-4


def _engine(tr):
    from loongx_amd.flux.engine import DiTEngine
    from loongx_amd.flux.weights import FluxConfig, pack_state_dict
    c = tr.config
    cfg = FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                     attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                     pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)
    return DiTEngine(pack_state_dict(tr.state_dict(), cfg, "cuda"), "cuda")


def test_engine_fp8_gemm_mode_against_reference_goldens():
    """The 4-block tiny model with every block GEMM on e4m3 operands (and, second line, fp8 attention too) vs the fp32 reference
    goldens: the error of the MODE, measured and bounded; the bf16 mode's bound on the same test is 2.5e-2."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    G = load("flux_tiny.npz")
    eng = _engine(tiny_transformer())
    d = "cuda"
    out = {}
    for name, mc in (("bf16", {}), ("gemm_fp8", {"gemm_fp8": True}), ("gemm_fp8+attn_fp8", {"gemm_fp8": True, "attn_fp8": True})):
        eng.set_conditioning(G["in_enc"].to(d), G["in_pooled"].to(d), G["in_guidance"].to(d), G["in_txt_ids"].to(d), G["in_img_ids"].to(d),
                             G["in_cond"].to(d), G["in_cond_ids"].to(d), model_config=mc)
        v = eng.forward(G["in_latents"].to(d), G["in_timestep"].to(d)).float().cpu().clone()
        out[name] = relerr(v, G["fwd_cond"])
        v2 = eng.forward(G["in_latents"].to(d), G["in_timestep"].to(d)).float().cpu()
        assert torch.equal(v, v2), name                                   # graph replay == first (eager-warmed) pass, deterministic
    print("FP8_MODE_RELERR " + json.dumps({k: round(v, 5) for k, v in out.items()}))
    assert out["bf16"] < 2.5e-2
    assert out["gemm_fp8"] < 0.15 and out["gemm_fp8+attn_fp8"] < 0.2
