"""The other BASELINE configs at their FULL shapes (configs[2..4]; configs[1] is what bench.py times).

  * configs[2] / [3]: batch 16 per GPU, all four modalities (EEG + fNIRS + PPG + motion), CS3 encoders + DGF fusion,
    512x512 -- through generate() with the full-size CS3 weights and a full-width (D=3072, 24 heads) 1+1-block DiT.
    Data-parallel sharding (configs[3]) is exactly "each rank runs a slice of the batch", so the property checked is the
    one that makes it correct: every sample of the batched run equals the run of that sample alone, bit for bit.
  * configs[4]: 1024x1024 (512 text + 4096 image + 4096 condition tokens, S = 8704) -- the attention kernel against an
    fp32 reference at that sequence length, and the engine's invariants (deterministic, batch-independent, condition
    masking decouples) at full width, in bf16; the fp8 attention path that config names (model_config attn_fp8) is measured at
    full depth by tests/test_fp8_gpu.py and at this sequence length by the attention tests below.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import relerr  # noqa: E402
from tests.test_kernels_gpu import BIASES, DEV, _attn_reference, _qkv_buffer, _segments, ops  # noqa: E402,F401

D, H, T = 3072, 24, 512
TOL_B4_ORACLE = 3.1e-3    # 2x the measured 1.53e-3 (4 steps, 1 + 1 full-width blocks, batch 4; round-3 audit)


def _model(dev="cuda"):
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig, synthetic_weights
    from loongx_amd.train.model import OminiModel, synthetic_cs3_state_dict
    pw = synthetic_weights(FluxConfig(num_layers=1, num_single_layers=1), dev, seed=0)
    return OminiModel.from_pipe(LxFluxPipeline(LxFluxTransformer(pw, dev)), synthetic_cs3_state_dict(0), {"union_cond_attn": True}, dev)


def test_generate_batch4_all_modalities_full_width_matches_the_oracle():
    """configs[2]'s composition at full width against the ORACLE (not only against itself): batch 4, all four modalities through the
    full-size CS3 encoders + DGF fusion (fuse_flag=True), a full-width (D = 3072, 24 heads, S = 2560) 1 + 1-block DiT, 4 denoise
    steps -- product generate() vs oracle.cs3.CS3DGF.brain_embeds + oracle.flux_ref.denoise_loop (fp32, DiT on this GPU)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.train.model import OminiModel
    from oracle import cs3 as ocs3
    from oracle import flux_modules as fm
    from oracle import flux_ref as fr
    from oracle.parity import build_pair
    dev = torch.device("cuda")
    tr, lx = build_pair(dev, 1, 1)
    torch.manual_seed(0)
    ref_cs3 = ocs3.CS3DGF(seed=0).eval()
    mc = {"union_cond_attn": True}
    model = OminiModel.from_pipe(LxFluxPipeline(lx), ref_cs3.state_dict(), mc, "cuda")
    B, hw, steps = 4, 32, 4
    N = hw * hw
    g = torch.Generator().manual_seed(21)
    r = lambda *s: torch.randn(*s, generator=g)
    lat, cond, pe, pooled = r(B, N, 64), r(B, N, 64), r(B, T, 4096) * 0.1, r(B, 768)
    eeg, fnirs, ppg, motion = r(B, 4, 4096), r(B, 6, 512), r(B, 4, 256), r(B, 6, 128)
    with torch.no_grad():
        rpe, rpool = ref_cs3.brain_embeds(pe, pooled, eeg, fnirs, ppg, motion, fuse_flag=True)
        ids = fm.prepare_latent_image_ids(hw, hw).to(dev)
        cids = ids.clone()
        cids[:, 2] -= hw
        want = fr.denoise_loop(tr, fm.FlowMatchEulerDiscreteScheduler(), lat.to(dev), rpe.to(dev), rpool.to(dev), torch.zeros(T, 3, device=dev),
                               ids, cond.to(dev), cids, num_inference_steps=steps)
    c = Condition("subject", latents=cond.cuda(), latent_hw=(hw, hw), position_delta=[0, -hw])
    out = generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=steps, latents=lat.cuda(),
                   prompt_embeds=pe.cuda(), pooled_prompt_embeds=pooled.cuda(), output_type="latent", model_config=mc, default_lora=True,
                   additional_condition1=eeg.cuda(), additional_condition2=fnirs.cuda(), additional_condition3=ppg.cuda(),
                   additional_condition4=motion.cuda(), use_brain_condition=True, fuse_flag=True).images
    assert out.shape == (B, N, 64)
    errs = [relerr(out[i].cpu(), want[i].cpu()) for i in range(B)]
    assert max(errs) < TOL_B4_ORACLE, errs
    # the brain side matters: the same call without the signals lands somewhere else
    plain = generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=steps, latents=lat.cuda(),
                     prompt_embeds=pe.cuda(), pooled_prompt_embeds=pooled.cuda(), output_type="latent", model_config=mc, default_lora=True,
                     use_brain_condition=False, fuse_flag=True).images
    assert relerr(plain.cpu(), want.cpu()) > 10 * TOL_B4_ORACLE


def test_generate_batch16_all_modalities_equals_single_runs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    model = _model()
    B, hw = 16, 32
    N = hw * hw
    g = torch.Generator(device="cuda").manual_seed(11)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = dict(lat=r(B, N, 64), cond=r(B, N, 64), pe=r(B, T, 4096) * 0.1, pooled=r(B, 768), eeg=r(B, 4, 4096), fnirs=r(B, 6, 512),
             ppg=r(B, 4, 256), motion=r(B, 6, 128))

    def run(sl):
        c = Condition("subject", latents=x["cond"][sl], latent_hw=(hw, hw), position_delta=[0, -hw])
        return generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"][sl],
                        prompt_embeds=x["pe"][sl], pooled_prompt_embeds=x["pooled"][sl], output_type="latent",
                        model_config=model.model_config, default_lora=True, additional_condition1=x["eeg"][sl],
                        additional_condition2=x["fnirs"][sl], additional_condition3=x["ppg"][sl], additional_condition4=x["motion"][sl],
                        use_brain_condition=True, fuse_flag=True).images.clone()

    eng = model.flux_pipe.transformer.engine
    # (1) launch plans that do not depend on the batch size: a data-parallel shard of any size reproduces the single-GPU
    #     batch BIT FOR BIT
    eng.pair_plan = False
    full = run(slice(None))
    assert full.shape == (B, N, 64) and torch.isfinite(full).all()
    assert torch.equal(full, run(slice(None)))                       # deterministic
    for i in (0, 7, 15):
        assert torch.equal(run(slice(i, i + 1))[0], full[i]), f"sample {i}"
    # (2) the default plans: a batch-1 step runs its N = 3072 long-K projections on the split-K pair kernel (one more fp32
    #     rounding per element), a batch-16 step has enough tiles not to -- equal within rounding, not bit for bit
    eng.pair_plan = True
    one = run(slice(7, 8))[0]
    assert torch.equal(one, run(slice(7, 8))[0])                     # still deterministic
    assert relerr(one.cpu(), full[7].cpu()) < 5e-3
    eng.check_status()
    # the brain conditioning is live: dropping the signals changes the result
    c = Condition("subject", latents=x["cond"][:1], latent_hw=(hw, hw), position_delta=[0, -hw])
    plain = generate(model, model.flux_pipe, conditions=[c], height=512, width=512, num_inference_steps=2, latents=x["lat"][:1],
                     prompt_embeds=x["pe"][:1], pooled_prompt_embeds=x["pooled"][:1], output_type="latent",
                     model_config=model.model_config, default_lora=True, use_brain_condition=False).images
    assert relerr(plain[0].cpu(), full[0].cpu()) > 1e-3


@pytest.mark.parametrize("mode", ["none", "cfactor"])
def test_attention_at_1024sq_sequence_length(ops, mode):
    """S = 512 + 4096 + 4096 = 8704 keys per query (136 KV tiles): long online-softmax chains, two heads."""
    lens, B, Hh = (512, 4096, 4096), 1, 2
    Dh = Hh * 128
    buf = _qkv_buffer(B, lens, Hh, seed=21)
    orig = buf.clone()
    row0, vt0, vt_len = _segments(B, lens)
    VT = torch.zeros(B, Hh, 128, vt_len, dtype=torch.bfloat16, device=DEV)
    for s, Ls in enumerate(lens):
        ops.qkv_prep(buf, q_col=2 * Dh, k_col=0, v_col=Dh, row0=row0[s], n_rows=B * Ls, rows_per_batch=Ls, H=Hh, wq=None, wk=None,
                     cos=None, sin=None, VT=VT, vt_pos0=vt0[s])
    bias = BIASES[mode]
    ops.attn_fwd(buf, buf, VT, buf, q_col=2 * Dh, k_col=0, o_col=2 * Dh, B=B, H=Hh, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=bias)
    ref, edges = _attn_reference(orig, B, Hh, lens, bias, 2 * Dh, 0, Dh)
    got = buf.float().cpu()
    for s, Ls in enumerate(lens):
        o = got[row0[s]: row0[s] + B * Ls, 2 * Dh: 3 * Dh].view(B, Ls, Hh, 128)
        assert relerr(o, ref[:, edges[s]:edges[s + 1]]) < 6e-3, f"segment {s}"


def test_attention_slow_max_growth_uses_deferred_rescale(ops):
    """Keys ordered so that the running row maximum creeps up by ~1 (log2 units) per tile over 40 tiles: with the deferred
    rescale (threshold 8) the kernel alternates between stale-max tiles and rescales; the result must match fp32 softmax."""
    B, Hh, Ls = 1, 1, 2560
    Dh = 128
    buf = _qkv_buffer(B, [Ls], Hh, seed=33)
    q = buf[:, 2 * Dh:3 * Dh].float()
    qn = q / q.norm(dim=-1, keepdim=True)
    # key j = direction of query 5 scaled so that score(q5, k_j) grows linearly with j  (scale = 1/sqrt(128), log2e folded in by the kernel)
    ramp = torch.linspace(0.0, 40.0 * math.log(2.0), Ls, device=DEV)             # natural-log units: +1 log2 unit per 64 keys
    k = qn[5].unsqueeze(0) * (ramp / (q[5].norm() / math.sqrt(128.0))).unsqueeze(1)
    buf[:, 0:Dh] = (k + 0.05 * torch.randn_like(k)).to(torch.bfloat16)
    orig = buf.clone()
    VT = torch.zeros(B, Hh, 128, Ls, dtype=torch.bfloat16, device=DEV)
    ops.qkv_prep(buf, q_col=2 * Dh, k_col=0, v_col=Dh, row0=0, n_rows=Ls, rows_per_batch=Ls, H=Hh, wq=None, wk=None, cos=None, sin=None,
                 VT=VT, vt_pos0=0)
    ops.attn_fwd(buf, buf, VT, buf, q_col=2 * Dh, k_col=0, o_col=2 * Dh, B=B, H=Hh, seg_row0=[0], seg_len=[Ls], seg_vt0=[0])
    ref, _ = _attn_reference(orig, B, Hh, (Ls,), [[0.0] * 3] * 3, 2 * Dh, 0, Dh)
    got = buf.float().cpu()[:, 2 * Dh:].view(1, Ls, 1, 128)
    assert relerr(got, ref) < 6e-3
    assert relerr(got[:, 5], ref[:, 5]) < 1e-2          # the row whose maximum ramps


def test_engine_invariants_at_1024sq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import flux_modules as fm
    model = _model()
    eng = model.flux_pipe.transformer.engine
    hw = 64
    N = hw * hw
    ids = fm.prepare_latent_image_ids(hw, hw).cuda()
    cids = ids.clone()
    cids[:, 2] -= hw
    g = torch.Generator(device="cuda").manual_seed(17)
    B = 2
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = dict(lat=r(B, N, 64), cond=r(B, N, 64), pe=r(B, T, 4096) * 0.1, pooled=r(B, 768))

    def fwd(sl, cond=None, mc=None):
        n = x["lat"][sl].shape[0]
        eng.set_conditioning(x["pe"][sl], x["pooled"][sl], torch.full((n,), 3.5, device="cuda"), torch.zeros(T, 3, device="cuda"), ids,
                             (x["cond"] if cond is None else cond)[sl], cids, model_config=mc or {})
        return eng.forward(x["lat"][sl], torch.full((n,), 0.7, device="cuda")).clone()

    v = fwd(slice(None))
    assert v.shape == (B, N, 64) and torch.isfinite(v).all()
    assert torch.equal(v, fwd(slice(None)))
    for i in range(B):                                # default plans (chosen by tile count): equal within rounding
        assert relerr(fwd(slice(i, i + 1))[0].cpu(), v[i].cpu()) < 5e-3
    eng.pair_plan = False                             # the batch-size-invariant plans: a shard reproduces the batch bit for bit
    v = fwd(slice(None))
    for i in range(B):
        assert torch.equal(fwd(slice(i, i + 1))[0], v[i])
    mc = {"union_cond_attn": False}
    other = torch.randn_like(x["cond"])
    assert torch.equal(fwd(slice(0, 1), mc=mc), fwd(slice(0, 1), cond=other, mc=mc))       # masked both ways: decoupled
    assert relerr(fwd(slice(0, 1), cond=other).cpu(), v[:1].cpu()) > 1e-3                   # union attention: coupled


def test_blocks_are_deterministic_under_back_to_back_load():
    """Race screen at full width: one double block and one (last) single block, 400 times back to back from the same residual
    stream, batch 4 (3840 attention workgroups per call, every GEMM plan). A missing barrier between the pipelined attention
    kernel's prologue and its first iteration (a lagging wave could read key tile 2 for its tile-0 scores) showed up exactly here,
    as different results for single 32-row groups in ~2 % of the runs, and nowhere in isolated kernel loops."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import flux_modules as fm
    model = _model()
    eng = model.flux_pipe.transformer.engine
    eng.pair_plan = False
    B, hw = 4, 32
    N = hw * hw
    g = torch.Generator(device="cuda").manual_seed(11)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    ids = fm.prepare_latent_image_ids(hw, hw).cuda()
    cids = ids.clone()
    cids[:, 2] -= hw
    eng.set_conditioning(r(B, T, 4096) * 0.1, r(B, 768), torch.full((B,), 3.5, device="cuda"), torch.zeros(T, 3, device="cuda"), ids, r(B, N, 64), cids,
                         c_t=0.0, model_config={"union_cond_attn": True})
    eng.embed_step_inputs(r(B, N, 64), torch.full((B,), 0.7, device="cuda"))
    X0 = eng.X.clone()

    def both():
        eng.X.copy_(X0)
        eng.double_block(0)
        eng.single_block(0, image_out_only=True)
        return eng.rows(eng.X, "img")
    ref = both().clone()
    assert torch.isfinite(ref).all()
    bad = sum(0 if torch.equal(both(), ref) else 1 for _ in range(400))
    assert bad == 0, f"{bad} of 400 back-to-back block executions differ"
