"""The reference-API mirrors (loongx_amd.flux.block / transformer / generate) on the GPU vs the reference-generated
goldens and the CPU oracle: these tests read like the reference's call sites."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cs3 as ocs3  # noqa: E402
from oracle import flux_modules as fm  # noqa: E402
from oracle import flux_ref as fr  # noqa: E402
from tests.helpers import load, relerr, tiny_transformer  # noqa: E402

# Tolerances = 2x what an MI355X measures against the reference-generated goldens (round-3 audit, LX_TEST_RECORD: gpurun_out/r03b):
TOL_ATTN = 1.2e-2     # attn_forward mirrors (bf16 q / k / v, one attention + projections): measured 5.2e-3 .. 5.9e-3
TOL_BLOCK = 3.6e-3    # one whole block on the fp32 residual stream: measured 1.5e-3 .. 1.8e-3
TOL_FWD = 8.4e-3      # 2 + 2 blocks, embedders, final layer: measured 4.2e-3
TOL_GEN = 3.2e-3      # 4-step denoise loops (the error of a velocity enters the latents times d sigma): measured 1.2e-3 .. 1.6e-3


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return load("flux_tiny.npz")


@pytest.fixture(scope="module")
def lx(G):
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    tr = tiny_transformer()
    c = tr.config
    cfg = FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, in_channels=64, joint_attention_dim=64,
                     pooled_projection_dim=32, guidance_embeds=True)
    return tr, LxFluxTransformer.from_state_dict(tr.state_dict(), cfg, "cuda")


def _ropes(tr, G):
    return tr.pos_embed(torch.cat([G["in_txt_ids"], G["in_img_ids"]], 0)), tr.pos_embed(G["in_cond_ids"])


def cu(t):
    return t.cuda()


MODES = {"default": ({}, None), "no_union": ({"union_cond_attn": False}, None), "independent": ({"independent_condition": True}, None),
         "cfactor_half": ({}, 0.5), "cfactor_two": ({}, 2.0), "latent_lora": ({"latent_lora": True}, None)}


@pytest.mark.parametrize("mode", list(MODES))
def test_attn_forward_mirror(G, lx, mode):
    from loongx_amd.flux.block import attn_forward
    tr, m = lx
    mc, cf = MODES[mode]
    main, cond = _ropes(tr, G)
    d, s = m.transformer_blocks[0].attn, m.single_transformer_blocks[0].attn
    s.text_len = 16
    try:
        if cf is not None:
            d.c_factor = s.c_factor = torch.ones(1, 1) * cf
        r = attn_forward(d, cu(G["hid"]), cu(G["enc"]), cu(G["cond"]), None, main, cond, mc)
        for got, key in zip(r, ("hid", "enc", "cond")):
            assert relerr(got.cpu(), G[f"attn_d_{mode}_{key}"]) < TOL_ATTN, key
        r = attn_forward(s, cu(torch.cat([G["enc"], G["hid"]], 1)), None, cu(G["cond"]), None, main, cond, mc)
        for got, key in zip(r, ("hid", "cond")):
            assert relerr(got.cpu(), G[f"attn_s_{mode}_{key}"]) < TOL_ATTN, key
    finally:
        for a in (d, s):
            if hasattr(a, "c_factor"):
                del a.c_factor


def test_attn_forward_mirror_nocond(G, lx):
    from loongx_amd.flux.block import attn_forward
    tr, m = lx
    main, _ = _ropes(tr, G)
    r = attn_forward(m.transformer_blocks[0].attn, cu(G["hid"]), cu(G["enc"]), None, None, main, None, {})
    assert len(r) == 2 and relerr(r[0].cpu(), G["attn_d_nocond_hid"]) < TOL_ATTN and relerr(r[1].cpu(), G["attn_d_nocond_enc"]) < TOL_ATTN
    s = m.single_transformer_blocks[0].attn
    s.text_len = 16
    r = attn_forward(s, cu(torch.cat([G["enc"], G["hid"]], 1)), None, None, None, main, None, {})
    assert relerr(r.cpu(), G["attn_s_nocond_hid"]) < TOL_ATTN
    with pytest.raises(NotImplementedError):
        attn_forward(s, cu(G["hid"]), None, None, torch.ones(1), main, None, {})


@pytest.mark.parametrize("name,mc", [("default", {}), ("add_cond", {"add_cond_attn": True})])
def test_block_forward_mirror(G, lx, name, mc):
    from loongx_amd.flux.block import block_forward
    tr, m = lx
    main, cond = _ropes(tr, G)
    e, h, c = block_forward(m.transformer_blocks[1], cu(G["hid"]), cu(G["enc"]), cu(G["cond"]), cu(G["temb"]), cu(G["ctemb"]), cond, main, mc)
    assert relerr(e.cpu(), G[f"block_{name}_enc"]) < TOL_BLOCK
    assert relerr(h.cpu(), G[f"block_{name}_hid"]) < TOL_BLOCK
    assert relerr(c.cpu(), G[f"block_{name}_cond"]) < TOL_BLOCK


@pytest.mark.parametrize("name", ["lora_off", "lora_half"])
def test_lora_controller_switches_against_the_reference(G, lx, name):
    """src/flux/lora_controller.py:5-75 -- `with enable_lora(modules, False)` (adapters off on every stream) and
    `with set_lora_scale(modules, 0.5)` around block_forward / single_block_forward, against goldens made by the reference's own
    context managers around its own block functions; the scale is restored on exit (the default goldens match again)."""
    from loongx_amd.flux.block import block_forward, single_block_forward
    from loongx_amd.flux.lora_controller import enable_lora, set_lora_scale
    tr, m = lx
    main, cond = _ropes(tr, G)
    dblk, sblk = m.transformer_blocks[1], m.single_transformer_blocks[1]
    sblk.text_len = 16
    ctx = (lambda h: enable_lora(h, False)) if name == "lora_off" else (lambda h: set_lora_scale(h, 0.5))
    hs = cu(torch.cat([G["enc"], G["hid"]], 1))
    with ctx([dblk, dblk.attn, object()]):                # non-engine objects are ignored, as the reference ignores non-PEFT modules
        e, h, c = block_forward(dblk, cu(G["hid"]), cu(G["enc"]), cu(G["cond"]), cu(G["temb"]), cu(G["ctemb"]), cond, main, {})
        h1, c1 = single_block_forward(sblk, hs, cu(G["temb"]), main, cu(G["cond"]), cu(G["ctemb"]), cond, {})
    for got, key in ((e, f"block_{name}_enc"), (h, f"block_{name}_hid"), (c, f"block_{name}_cond"), (h1, f"single_{name}_hid"), (c1, f"single_{name}_cond")):
        assert relerr(got.cpu(), G[key]) < TOL_BLOCK, key
    # the switch matters (the condition stream moves by 3-8 % between the settings) and is restored on exit
    assert relerr(c.cpu(), G["block_default_cond"]) > 2 * TOL_BLOCK
    assert m.engine.lora_scale == 1.0
    e, h, c = block_forward(dblk, cu(G["hid"]), cu(G["enc"]), cu(G["cond"]), cu(G["temb"]), cu(G["ctemb"]), cond, main, {})
    assert relerr(c.cpu(), G["block_default_cond"]) < TOL_BLOCK
    with enable_lora([dblk], True):                       # activated=True: a no-op, as in the reference
        assert m.engine.lora_scale == 1.0


def test_block_forward_mirror_nocond(G, lx):
    from loongx_amd.flux.block import block_forward
    tr, m = lx
    main, _ = _ropes(tr, G)
    e, h, c = block_forward(m.transformer_blocks[1], cu(G["hid"]), cu(G["enc"]), None, cu(G["temb"]), None, None, main, {})
    assert c is None and relerr(e.cpu(), G["block_nocond_enc"]) < TOL_BLOCK and relerr(h.cpu(), G["block_nocond_hid"]) < TOL_BLOCK


def test_single_block_forward_mirror(G, lx):
    from loongx_amd.flux.block import single_block_forward
    tr, m = lx
    main, cond = _ropes(tr, G)
    blk = m.single_transformer_blocks[1]
    blk.text_len = 16
    hs = cu(torch.cat([G["enc"], G["hid"]], 1))
    h, c = single_block_forward(blk, hs, cu(G["temb"]), main, cu(G["cond"]), cu(G["ctemb"]), cond, {})
    assert relerr(h.cpu(), G["single_hid"]) < TOL_BLOCK and relerr(c.cpu(), G["single_cond"]) < TOL_BLOCK
    h2 = single_block_forward(blk, hs, cu(G["temb"]), main)
    assert relerr(h2.cpu(), G["single_nocond_hid"]) < TOL_BLOCK


def test_tranformer_forward_mirror(G, lx):
    from loongx_amd.flux.transformer import tranformer_forward
    tr, m = lx
    kw = dict(hidden_states=cu(G["in_latents"]), encoder_hidden_states=cu(G["in_enc"]), pooled_projections=cu(G["in_pooled"]),
              timestep=cu(G["in_timestep"]), img_ids=cu(G["in_img_ids"]), txt_ids=cu(G["in_txt_ids"]), guidance=cu(G["in_guidance"]))
    out = tranformer_forward(m, cu(G["in_cond"]), cu(G["in_cond_ids"]), None, {}, return_dict=False, **kw)
    assert isinstance(out, tuple) and relerr(out[0].cpu(), G["fwd_cond"]) < TOL_FWD
    out = tranformer_forward(m, None, None, None, {}, **kw)
    assert relerr(out.sample.cpu(), G["fwd_nocond"]) < TOL_FWD
    with pytest.raises(NotImplementedError):
        tranformer_forward(m, None, None, None, {}, controlnet_block_samples=[1], **kw)


# ------------------------------------------------------------------------------------------ generate()
def _mk_model(tr):
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.flux.weights import FluxConfig
    from loongx_amd.train.model import OminiModel
    cfg = FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, in_channels=64, joint_attention_dim=4096,
                     pooled_projection_dim=768, guidance_embeds=True)
    torch.manual_seed(0)
    ref_cs3 = ocs3.CS3DGF(seed=0).eval()
    lxtr = LxFluxTransformer.from_state_dict(tr.state_dict(), cfg, "cuda")
    return ref_cs3, OminiModel.from_pipe(LxFluxPipeline(lxtr), ref_cs3.state_dict(), {}, "cuda")


def test_generate_matches_oracle_loop():
    """4-step 64x64 edit with EEG+PPG / fNIRS+Motion conditioning, latents in/out (no T5, no VAE)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    tr = fm.FluxTransformer2DModel(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=4096,
                                   pooled_dim=768, guidance_embeds=True, lora=True)
    fm.init_synthetic_(tr, seed=4, std=0.03, bias_std=0.02, norm_jitter=0.1)
    tr.eval()
    ref_cs3, model = _mk_model(tr)
    g = torch.Generator().manual_seed(3)
    B, hw = 1, 4
    lat = torch.randn(B, hw * hw, 64, generator=g)
    cond = torch.randn(B, hw * hw, 64, generator=g)
    pe, pooled = torch.randn(B, 512, 4096, generator=g) * 0.1, torch.randn(B, 768, generator=g)
    eeg, ppg = torch.randn(4, 3000, generator=g), torch.randn(4, 256, generator=g)
    fnirs, motion = torch.randn(6, 600, generator=g), torch.randn(6, 128, generator=g)
    allsig = dict(eeg=eeg, ppg=ppg, fnirs=fnirs, motion=motion)
    # (fuse_flag, signals, brain_replace): "both" = the reference's literal rule and the default (EEG alone leaves the text
    # embeddings untouched, generate.py:252-255); "per_stream" = the opt-in extension BASELINE configs[1] runs with
    outs = {}
    for fuse_flag, sig, rule in ((False, dict(eeg=eeg), "per_stream"), (False, dict(eeg=eeg), "both"), (False, allsig, "both"),
                                 (True, allsig, "both"), (False, {}, "both")):
        with torch.no_grad():
            s = {k: v.unsqueeze(0) for k, v in sig.items()}
            rpe, rpool = ref_cs3.brain_embeds(pe, pooled, s.get("eeg"), s.get("fnirs"), s.get("ppg"), s.get("motion"), fuse_flag=fuse_flag,
                                              per_stream=rule == "per_stream")
            ids = fm.prepare_latent_image_ids(hw, hw)
            cids = ids.clone()
            cids[:, 2] -= hw
            want = fr.denoise_loop(tr, fm.FlowMatchEulerDiscreteScheduler(), lat, rpe, rpool, torch.zeros(512, 3), ids, cond, cids,
                                   num_inference_steps=4)
        c = Condition("subject", latents=cond.cuda(), latent_hw=(hw, hw), position_delta=[0, -hw])
        out = generate(model, model.flux_pipe, conditions=[c], height=hw * 16, width=hw * 16, num_inference_steps=4, latents=lat.cuda(),
                       prompt_embeds=pe.cuda(), pooled_prompt_embeds=pooled.cuda(), output_type="latent", model_config={},
                       default_lora=True, additional_condition1=sig.get("eeg"), additional_condition2=sig.get("fnirs"),
                       additional_condition3=sig.get("ppg"), additional_condition4=sig.get("motion"),
                       use_brain_condition=bool(sig), fuse_flag=fuse_flag, **({} if rule == "both" else {"brain_replace": rule}))
        assert relerr(out.images.cpu(), want) < TOL_GEN, (fuse_flag, list(sig), rule)
        outs[(fuse_flag, tuple(sig), rule)] = out.images.clone()
    # the literal rule ignores a lone EEG (== no signals at all); the per-stream rule does not
    assert torch.equal(outs[(False, ("eeg",), "both")], outs[(False, (), "both")])
    assert relerr(outs[(False, ("eeg",), "per_stream")].cpu(), outs[(False, (), "both")].cpu()) > 1e-3


@pytest.mark.parametrize("case", [c[0] for c in __import__("oracle.ducks", fromlist=["x"]).generate_cases()])
def test_generate_matches_the_reference_generate_goldens(case):
    """The product's generate() vs goldens made by the reference's REAL generate() (src/flux/generate.py:72-394) driven through
    a duck-typed pipeline / model (oracle/make_goldens.py::gold_generate): same call, same kwargs, the condition as an image
    object through Condition.encode + encode_images (duck VAE), default reference replacement rule, DUAN fusion,
    condition_scale -> c_factor. Documented delta Q1 (signals reach the encoders as [B,C,L]) is part of the golden."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import ducks
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    from loongx_amd.flux.pipeline import LxFluxPipeline
    G = load("generate_tiny.npz")
    name, fuse_flag, use, cscale = next(c for c in ducks.generate_cases() if c[0] == case)
    _, model = _mk_model(ducks.generate_transformer())
    pipe = LxFluxPipeline(model.transformer, vae=ducks.DuckVAE(), image_processor=ducks.DuckImageProcessor())
    hw = 4
    cond = Condition(condition_type="subject", condition=ducks.DuckImage(hw * 16, hw * 16, seed=5))
    sig = {k: (G["in_" + k].cuda() if k in use else None) for k in ("eeg", "fnirs", "ppg", "motion")}
    out = generate(model, pipe, conditions=[cond], height=hw * 16, width=hw * 16, num_inference_steps=4, latents=G["in_lat"].cuda(),
                   prompt_embeds=G["in_pe"].cuda(), pooled_prompt_embeds=G["in_pooled"].cuda(), output_type="latent", model_config={},
                   default_lora=True, condition_scale=cscale, additional_condition1=sig["eeg"], additional_condition2=sig["fnirs"],
                   additional_condition3=sig["ppg"], additional_condition4=sig["motion"], use_brain_condition=bool(use),
                   fuse_flag=fuse_flag, return_dict=False)
    assert isinstance(out, tuple)
    assert relerr(out[0].cpu(), G[f"gen_{case}"]) < TOL_GEN
    assert cond.position_delta == [0, -hw]                      # default subject delta, written back (condition.py:126-127)
    assert model.transformer.c_factor is None                   # removed again on exit (generate.py:384-388)
    assert torch.equal(pipe.scheduler.timesteps.cpu(), G["sched_timesteps"]) and torch.equal(pipe.scheduler.sigmas.cpu(), G["sched_sigmas"])


def test_scheduler_and_latent_utils_match_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd.flux.pipeline import FlowMatchEulerDiscreteScheduler, LxFluxPipeline, calculate_shift
    a, b = FlowMatchEulerDiscreteScheduler(), fm.FlowMatchEulerDiscreteScheduler()
    sig = np.linspace(1.0, 1 / 28, 28)
    mu = calculate_shift(1024, 256, 4096, 0.5, 1.15)
    assert mu == fm.calculate_shift(1024, 256, 4096, 0.5, 1.15)
    a.set_timesteps(sigmas=sig, mu=mu, device="cuda"); b.set_timesteps(sigmas=sig, mu=mu)
    assert torch.equal(a.timesteps.cpu(), b.timesteps) and torch.equal(a.sigmas.cpu(), b.sigmas)
    x, v = torch.randn(2, 16, 64), torch.randn(2, 16, 64)
    for i in range(3):
        xa = a.step(v.cuda(), a.timesteps[i], x.cuda())[0].cpu()
        xb = b.step(v, b.timesteps[i], x)[0]
        assert torch.allclose(xa, xb, atol=1e-6)
        x = xb
    z = torch.randn(2, 16, 8, 8)
    p = LxFluxPipeline._pack_latents(z, 2, 16, 8, 8)
    assert torch.equal(p, fm.pack_latents(z)) and torch.equal(LxFluxPipeline._unpack_latents(p, 64, 64, 16), z)
    assert torch.equal(LxFluxPipeline._prepare_latent_image_ids(1, 8, 8, "cpu", torch.float32), fm.prepare_latent_image_ids(4, 4))


def test_generate_edge_cases_batch_nonsquare_callback_nocondition_errors():
    """generate() surface beyond the golden cases: a batch of two prompts, a non-square 64x96 edit (24 image tokens: ragged against
    every tile size), no condition at all, `callback_on_step_end` rewriting the latents, the latent noise drawn from `generator`,
    `num_images_per_prompt`, and the input checks of the reference pipeline (generate.py:97-106)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import ducks
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    tr = ducks.generate_transformer()
    _, model = _mk_model(tr)
    pipe = model.flux_pipe
    g = torch.Generator().manual_seed(12)
    B, H2, W2 = 2, 4, 6                                                 # 64 x 96 pixels -> 4 x 6 packed grid
    lat = torch.randn(B, H2 * W2, 64, generator=g)
    cond = torch.randn(B, H2 * W2, 64, generator=g)
    pe, pooled = torch.randn(B, 512, 4096, generator=g) * 0.1, torch.randn(B, 768, generator=g)
    ids = fm.prepare_latent_image_ids(H2, W2)
    cids = ids.clone()
    cids[:, 2] -= W2
    with torch.no_grad():
        want = fr.denoise_loop(tr, fm.FlowMatchEulerDiscreteScheduler(), lat, pe, pooled, torch.zeros(512, 3), ids, cond, cids, num_inference_steps=3)
        want_nc = fr.denoise_loop(tr, fm.FlowMatchEulerDiscreteScheduler(), lat, pe, pooled, torch.zeros(512, 3), ids, None, None, num_inference_steps=3)
    kw = dict(height=H2 * 16, width=W2 * 16, num_inference_steps=3, prompt_embeds=pe.cuda(), pooled_prompt_embeds=pooled.cuda(),
              output_type="latent", model_config={}, default_lora=True, use_brain_condition=False)
    c = Condition("subject", latents=cond.cuda(), latent_hw=(H2, W2), position_delta=[0, -W2])
    out = generate(model, pipe, conditions=[c], latents=lat.cuda(), **kw)
    assert out.images.shape == (B, H2 * W2, 64) and relerr(out.images.cpu(), want) < TOL_GEN
    out_nc = generate(model, pipe, conditions=None, latents=lat.cuda(), **kw)
    assert relerr(out_nc.images.cpu(), want_nc) < TOL_GEN and relerr(out_nc.images.cpu(), want) > 1e-3
    # callback: sees every step, may replace the latents (here: zero them after the last step)
    seen = []

    def cb(pipe_, i, t, kwargs):
        seen.append((i, float(t), tuple(kwargs["latents"].shape)))
        return {"latents": kwargs["latents"] * 0} if i == 2 else {}
    z = generate(model, pipe, conditions=[c], latents=lat.cuda(), callback_on_step_end=cb, **kw).images
    assert [s[0] for s in seen] == [0, 1, 2] and seen[0][1] > seen[2][1] and float(z.abs().max()) == 0.0
    # noise from the generator: reproducible, and num_images_per_prompt multiplies the batch
    kw1 = dict(kw, prompt_embeds=pe[:1].cuda(), pooled_prompt_embeds=pooled[:1].cuda())
    a = generate(model, pipe, conditions=None, generator=torch.Generator(device="cuda").manual_seed(3), **kw1).images
    b = generate(model, pipe, conditions=None, generator=torch.Generator(device="cuda").manual_seed(3), **kw1).images
    assert torch.equal(a, b) and a.shape == (1, H2 * W2, 64)
    two = generate(model, pipe, conditions=None, num_images_per_prompt=2, generator=torch.Generator(device="cuda").manual_seed(3), **kw1).images
    assert two.shape == (2, H2 * W2, 64) and not torch.equal(two[0], two[1])
    # input checks (diffusers FluxPipeline.check_inputs, called at generate.py:97-106)
    with pytest.raises(ValueError):
        generate(model, pipe, conditions=None, latents=lat.cuda(), **dict(kw, height=66))
    with pytest.raises(ValueError):
        generate(model, pipe, conditions=None, latents=lat.cuda(), prompt="x", **kw)
    with pytest.raises(ValueError):
        generate(model, pipe, conditions=None, latents=lat.cuda(), **{k: v for k, v in kw.items() if k != "pooled_prompt_embeds"})
    with pytest.raises(AssertionError):
        generate(model, pipe, conditions=[c, c], latents=lat.cuda(), **kw)
    with pytest.raises(NotImplementedError):                            # no text encoder plugged in: a prompt string cannot be encoded
        generate(model, pipe, conditions=None, latents=lat.cuda(), prompt="make it red", **{k: v for k, v in kw.items() if "embeds" not in k})
