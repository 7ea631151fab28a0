"""DiT engine (HIP kernels composed by loongx_amd.flux.engine) vs the CPU oracle and the reference-generated goldens."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_ref as fr  # noqa: E402
from tests.helpers import load, relerr, tiny_transformer  # noqa: E402

# bf16 GEMM operands / bf16 attention against an fp32 oracle on a 4-block model
TOL = 9e-3      # 2x the measured 4.0e-3 .. 4.5e-3 of a 2 + 2-block forward against the reference goldens (round-3 audit)


def _engine(tr, lora_scale=1.0):
    from loongx_amd.flux.engine import DiTEngine
    from loongx_amd.flux.weights import FluxConfig, pack_state_dict
    c = tr.config
    cfg = FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                     attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                     pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)
    return DiTEngine(pack_state_dict(tr.state_dict(), cfg, "cuda", lora_scale=lora_scale), "cuda")


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return load("flux_tiny.npz")


def _run(eng, G, cond=True, c_t=0.0, guidance=True, model_config=None, c_factor=None):
    d = "cuda"
    eng.set_conditioning(G["in_enc"].to(d), G["in_pooled"].to(d), G["in_guidance"].to(d) if guidance else None, G["in_txt_ids"].to(d),
                         G["in_img_ids"].to(d), G["in_cond"].to(d) if cond else None, G["in_cond_ids"].to(d) if cond else None,
                         c_t=c_t, model_config=model_config or {}, c_factor=c_factor)
    return eng.forward(G["in_latents"].to(d), G["in_timestep"].to(d)).float().cpu().clone()


def test_forward_matches_reference_goldens(G):
    eng = _engine(tiny_transformer())
    assert relerr(_run(eng, G), G["fwd_cond"]) < TOL
    assert relerr(_run(eng, G, cond=False), G["fwd_nocond"]) < TOL
    assert relerr(_run(eng, G, c_t=0.25), G["fwd_cond_ct025"]) < TOL
    assert relerr(_run(eng, G), G["fwd_cond"]) < TOL      # shape/conditioning switches leave no stale state


def test_bounded_score_attention_is_decided_per_layer(G, monkeypatch):
    """The engine runs the no-running-maximum attention kernel (engine._setup_nomax) only on layers whose norm_q / norm_k weights keep
    16.33 max|norm_q| max|norm_k| + bias under 100; a layer outside the bound keeps the max-tracking kernel. Either way the forward
    agrees with the fp32 oracle, and with the engine that never uses the bounded kernel (LX_ATTN_NOMAX=0)."""
    tr = tiny_transformer()
    blk = tr.transformer_blocks[1]
    with torch.no_grad():
        blk.attn.norm_q.weight.mul_(4.0)           # 16.33 x 4 x 2.5 = 163 > 100: this layer must fall back
        blk.attn.norm_k.weight.mul_(2.5)
        ref = fr.tranformer_forward(tr, G["in_cond"], G["in_cond_ids"], None, {}, hidden_states=G["in_latents"],
                                    encoder_hidden_states=G["in_enc"], pooled_projections=G["in_pooled"], timestep=G["in_timestep"],
                                    img_ids=G["in_img_ids"], txt_ids=G["in_txt_ids"], guidance=G["in_guidance"])[0]
    eng = _engine(tr)
    got = _run(eng, G)
    w = eng.w.t
    assert eng.attn_nomax and not eng._layer_nomax(w["d1.wq"]) and eng._layer_nomax(w["d0.wq"]) and eng._layer_nomax(w["s0.wq"])
    assert eng._qn(w["d1.wq_txt"], w["d1.wq"]) is w["d1.wq_txt"] and eng._qn(w["d0.wq"], w["d0.wq"]) is not w["d0.wq"]
    assert relerr(got, ref) < TOL
    monkeypatch.setenv("LX_ATTN_NOMAX", "0")
    eng0 = _engine(tr)
    got0 = _run(eng0, G)
    assert not eng0.attn_nomax and relerr(got0, ref) < TOL and relerr(got, got0) < 6e-3
    # a bias large enough to use up the room (c_factor = e^70: log2 units 101) switches every layer back
    monkeypatch.delenv("LX_ATTN_NOMAX")
    _run(eng, G, c_factor=2.5e30)
    assert not eng.attn_nomax


def test_forward_no_guidance(G):
    eng = _engine(tiny_transformer(seed=3, guidance_embeds=False))
    assert relerr(_run(eng, G, guidance=False), G["fwd_noguidance_seed3"]) < TOL


@pytest.mark.parametrize("mc,cf", [({"union_cond_attn": False}, None), ({"independent_condition": True}, None), ({}, 0.5), ({}, 2.0),
                                   ({"latent_lora": True}, None), ({"add_cond_attn": True}, None)])
def test_forward_model_config_modes(G, mc, cf):
    """Mask modes / c_factor / latent_lora / add_cond_attn vs the oracle restatement (itself pinned to the goldens)."""
    tr = tiny_transformer()
    if cf is not None:
        for m in tr.modules():
            if hasattr(m, "to_q"):
                m.c_factor = torch.ones(1, 1) * cf
    with torch.no_grad():
        ref = fr.tranformer_forward(tr, G["in_cond"], G["in_cond_ids"], None, mc, hidden_states=G["in_latents"],
                                    encoder_hidden_states=G["in_enc"], pooled_projections=G["in_pooled"], timestep=G["in_timestep"],
                                    img_ids=G["in_img_ids"], txt_ids=G["in_txt_ids"], guidance=G["in_guidance"])[0]
    eng = _engine(tr)
    got = _run(eng, G, model_config=mc, c_factor=cf)
    assert relerr(got, ref) < TOL
    if mc or cf:
        assert relerr(got, G["fwd_cond"]) > 1e-3    # the mode actually changes the result


def test_forward_is_deterministic(G):
    eng = _engine(tiny_transformer())
    a, b = _run(eng, G), _run(eng, G)
    assert torch.equal(a, b)


@pytest.mark.parametrize("mc", [{}, {"latent_lora": True}])
def test_prepared_schedule_matches_per_step(G, mc):
    """prepare_schedule(): all steps' modulation vectors from ONE pass over the modulation weights (MFMA GEMM on bf16 hi/lo
    halves of silu(temb): fp32-class accuracy) against the per-step evaluation (weight-streaming GEMV on fp32 activations),
    with and without LoRA on the image stream; the prepared path itself is deterministic."""
    d = "cuda"
    eng = _engine(tiny_transformer())
    ts = [1.0, 0.8731, 0.5, 0.25, 0.0357]
    B = G["in_latents"].shape[0]
    per_step = []
    for t in ts:
        tt = torch.full((B,), t, device=d)
        _run(eng, G, model_config=mc)                                 # (re)conditioning drops any prepared schedule
        assert eng.sched is None
        per_step.append(eng.forward(G["in_latents"].to(d), tt).clone())
    _run(eng, G, model_config=mc)
    eng.prepare_schedule(torch.tensor(ts))
    for i, t in enumerate(ts):                                        # the table itself against the per-step kernels
        eng.t1000.copy_(torch.full((B,), t * 1000.0, device=d))
        eng._time_text_embed(eng.t1000, eng.temb, eng.temb_base)
        eng._compute_mods(eng.temb, eng.mods, lora=eng.latent_lora)
        assert relerr(eng.sched[1][i].cpu(), eng.mods.cpu()) < 1e-5, f"modulation table, step {i}"
    for i, t in enumerate(ts):
        tt = torch.full((B,), t, device=d)
        got = eng.forward(G["in_latents"].to(d), tt, step_index=i).clone()
        # a 2.5e-6 difference in the modulations flips occasional bf16 roundings downstream: equal within bf16 noise
        assert relerr(got.cpu(), per_step[i].cpu()) < 5e-3, f"step {i}"
        assert torch.equal(got, eng.forward(G["in_latents"].to(d), tt, step_index=i))
    with pytest.raises(IndexError):
        eng.forward(G["in_latents"].to(d), tt, step_index=len(ts))
    # conditioning change invalidates the table: step_index is then ignored and the per-step path runs
    _run(eng, G, model_config=mc)
    got = eng.forward(G["in_latents"].to(d), torch.full((B,), ts[2], device=d), step_index=2)
    assert torch.equal(got, per_step[2])


def test_load_lora_from_safetensors(G, tmp_path):
    """model.py:463-477 / inference.py:43-44: base weights first, then `load_lora(dir)` with a diffusers-format
    `pytorch_lora_weights.safetensors` (`transformer.<module>.lora_A.weight`): the result must equal packing the PEFT-wrapped
    state dict directly (bit for bit) and reproduce the reference golden; an unknown module name is an error."""
    from safetensors.torch import save_file
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    tr = tiny_transformer()
    sd = tr.state_dict()
    base = {k.replace(".base_layer.", "."): v for k, v in sd.items() if ".lora_" not in k}
    lora = {"transformer." + k.replace(".default.", "."): v.contiguous() for k, v in sd.items() if ".lora_" in k}
    assert len(lora) == 50
    save_file(lora, str(tmp_path / "pytorch_lora_weights.safetensors"))
    c = tr.config
    cfg = FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                     attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                     pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)
    lxt = LxFluxTransformer.from_state_dict(base, cfg, "cuda")
    pipe = LxFluxPipeline(lxt)
    eng = lxt.engine
    no_lora = _run(eng, G)
    assert relerr(no_lora, G["fwd_cond"]) > 1e-3                      # adapters matter for the condition stream
    assert pipe.load_lora_weights(str(tmp_path)) == 25
    got = _run(eng, G)
    assert relerr(got, G["fwd_cond"]) < TOL
    assert torch.equal(got, _run(_engine(tr), G))                      # same tensors as packing the wrapped state dict
    bad = dict(lora)
    bad["transformer.transformer_blocks.9.attn.to_q.lora_A.weight"] = bad["transformer.x_embedder.lora_A.weight"].clone()
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(KeyError):
        pipe.load_lora_weights(str(tmp_path / "bad.safetensors"))
    with pytest.raises(FileNotFoundError):
        pipe.load_lora_weights(str(tmp_path / "nowhere"))


@pytest.mark.parametrize("mc", [{}, {"latent_lora": True}, {"union_cond_attn": False}])
def test_forward_with_fused_qkv_epilogue(mc):
    """Streams of 32 / 64 / 64 tokens take the LX_EPI_QKV projection epilogue (RMSNorm + RoPE + V^T inside the GEMM, no qkv_prep
    launch): against the fp32 oracle, and against the same engine with qkv_epilogue = False (the two-pass path the goldens cover).
    The last single block exercises the kv-only (N = 2D) launch of the text / condition streams."""
    from oracle import flux_modules as fm
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
    outs = {}
    for fused in ("1", "0"):
        eng = _engine(tr)
        eng.qkv_epilogue = fused == "1"
        eng.set_conditioning(kw["encoder_hidden_states"].to(d), kw["pooled_projections"].to(d), kw["guidance"].to(d), kw["txt_ids"].to(d),
                             kw["img_ids"].to(d), cond.to(d), cids.to(d), c_t=0.0, model_config=mc)
        assert eng.qkv_fused == (fused == "1")
        a = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()
        b = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()      # graph replay
        assert torch.equal(a, b)
        outs[fused] = a
    e1, e0 = relerr(outs["1"], want), relerr(outs["0"], want)
    assert e1 < TOL and e0 < TOL, (e1, e0)
    assert relerr(outs["1"], outs["0"]) < 1e-2


@pytest.mark.parametrize("operands", ["bf16", "fp16"])
@pytest.mark.parametrize("mc", [{}, {"latent_lora": True}, {"independent_condition": True}])
def test_lora_down_inside_the_layernorm_launch_matches_the_separate_launches(mc, operands):
    """engine.ln_lora (default on): the LayerNorm launches also compute the adapter down-projection of the Linear that reads them
    (lx_ln_modulate_lora[_f16]_segs, MFMA inside the launch) -- q/k/v and ff1 of the double blocks, the fused projection of the single
    blocks; with latent_lora the image (and in single blocks the text) rows too. Against the fp32 oracle, and against the same engine
    with the separate lx_lora_down launches: the same operands, another split of K in the fp32 sums."""
    from oracle import flux_modules as fm
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
    outs = {}
    from loongx_amd import ops
    for fusedln in (True, False):
        eng = _engine(tr)
        eng.ln_lora = fusedln
        eng.set_conditioning(kw["encoder_hidden_states"].to(d), kw["pooled_projections"].to(d), kw["guidance"].to(d), kw["txt_ids"].to(d),
                             kw["img_ids"].to(d), cond.to(d), cids.to(d), c_t=0.0, model_config=dict(mc, operands=operands))
        calls = []
        real = ops.lora_down
        ops.lora_down = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            eng.use_graph = False
            a = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()
        finally:
            ops.lora_down = real
        eng.use_graph = True
        b = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()      # captured
        c = eng.forward(kw["hidden_states"].to(d), kw["timestep"].to(d)).float().cpu().clone()      # replayed
        assert torch.equal(a, b) and torch.equal(b, c)
        outs[fusedln] = (a, len(calls))
    (a1, n1), (a0, n0) = outs[True], outs[False]
    assert n0 - n1 == 4, (n1, n0)                             # the launches behind a LayerNorm are gone: q/k/v of the 2 double blocks, the fused
                                                              # projection of the 2 single blocks (to_out / ff.net.2 / proj_out read GEMM outputs)
    assert relerr(a1, want) < TOL and relerr(a0, want) < TOL
    assert relerr(a1, a0) < (2e-3 if operands == "bf16" else 4e-4)       # fp32 sums in another order, then 16-bit operand roundings downstream


@pytest.mark.parametrize("mc", [{"independent_condition": True}, {"union_cond_attn": False}, {"independent_condition": True, "latent_lora": True}])
def test_step_invariant_condition_stream_is_cached(monkeypatch, mc):
    """independent_condition / union_cond_attn = False (block.py:106-120): the condition queries see only condition keys and c_t is
    fixed, so the condition stream repeats itself every denoise step. The engine computes it in the first forward of a
    conditioning, keeps its keys / V^T per layer and runs only the text and image rows afterwards. Three steps with different
    latents and timesteps: against the fp32 oracle (which recomputes everything every step) and against the engine with
    LX_COND_CACHE=0; a new conditioning must invalidate the cache."""
    from oracle import flux_modules as fm
    tr = tiny_transformer(seed=5)
    g = torch.Generator().manual_seed(11)
    B, T, hw = 2, 32, 8
    N = hw * hw
    enc, pooled = torch.randn(B, T, 64, generator=g) * 0.5, torch.randn(B, 32, generator=g)
    ids, tids = fm.prepare_latent_image_ids(hw, hw), torch.zeros(T, 3)
    cids = ids.clone()
    cids[:, 2] -= hw
    conds = [torch.randn(B, N, 64, generator=g) for _ in range(2)]
    lats = [torch.randn(B, N, 64, generator=g) for _ in range(3)]
    ts = [torch.tensor([0.9, 0.8]), torch.tensor([0.55, 0.5]), torch.tensor([0.2, 0.1])]
    guid = torch.full((B,), 3.5)
    def oracle(cond, lat, t):
        with torch.no_grad():
            return fr.tranformer_forward(tr, cond, cids, None, mc, hidden_states=lat, encoder_hidden_states=enc, pooled_projections=pooled,
                                         timestep=t, img_ids=ids, txt_ids=tids, guidance=guid)[0]
    d = "cuda"
    res = {}
    for cache in ("1", "0"):
        monkeypatch.setenv("LX_COND_CACHE", cache)
        eng = _engine(tr)
        outs = []
        for ci, cond in enumerate(conds):
            eng.set_conditioning(enc.to(d), pooled.to(d), guid.to(d), tids.to(d), ids.to(d), cond.to(d), cids.to(d), c_t=0.0, model_config=mc)
            for k in range(3):
                assert eng.cond_cached == (cache == "1" and k > 0)
                outs.append(eng.forward(lats[k].to(d), ts[k].to(d)).float().cpu().clone())
                assert eng.cond_cache == (cache == "1")
        res[cache] = outs
    for ci, cond in enumerate(conds):
        for k in range(3):
            want = oracle(cond, lats[k], ts[k])
            e1, e0 = relerr(res["1"][ci * 3 + k], want), relerr(res["0"][ci * 3 + k], want)
            assert e1 < TOL and e0 < TOL, (ci, k, e1, e0)
            assert relerr(res["1"][ci * 3 + k], res["0"][ci * 3 + k]) < 5e-3
    if mc.get("union_cond_attn", True):                        # (without union attention the image never sees the condition at all)
        assert relerr(res["1"][0], res["1"][3]) > 1e-3        # the two conditionings differ: the cache really was refreshed


def test_condition_cache_is_off_when_the_condition_stream_sees_the_image(G):
    eng = _engine(tiny_transformer())
    _run(eng, G)                                              # default union attention: nothing to cache
    assert not eng.cond_cache and not eng.cond_cached and eng.KC is None


