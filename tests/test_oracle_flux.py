"""Pins oracle/flux_ref.py (restatement of src/flux/block.py + transformer.py) against golden vectors
produced by the REAL reference functions (oracle/make_goldens.py).  CPU only."""
import pytest
import torch

from oracle import flux_modules as fm
from oracle import flux_ref as fr
from tests.helpers import load, relerr, tiny_transformer

TOL = 2e-6  # fp32 re-association only


@pytest.fixture(scope="module")
def G():
    return load("flux_tiny.npz")


@pytest.fixture(scope="module")
def tr():
    return tiny_transformer()


def _ropes(tr, G):
    main = tr.pos_embed(torch.cat([G["in_txt_ids"], G["in_img_ids"]], 0))
    cond = tr.pos_embed(G["in_cond_ids"])
    return main, cond


MODES = {"default": ({}, None), "no_union": ({"union_cond_attn": False}, None),
         "independent": ({"independent_condition": True}, None), "cfactor_half": ({}, 0.5),
         "cfactor_two": ({}, 2.0), "latent_lora": ({"latent_lora": True}, None)}


@pytest.mark.parametrize("mode", list(MODES))
def test_attn_forward_modes(G, tr, mode):
    mc, cf = MODES[mode]
    main, cond = _ropes(tr, G)
    d, s = tr.transformer_blocks[0].attn, tr.single_transformer_blocks[0].attn
    try:
        if cf is not None:
            d.c_factor = s.c_factor = torch.ones(1, 1) * cf
        with torch.no_grad():
            r = fr.attn_forward(d, G["hid"], G["enc"], G["cond"], None, main, cond, mc)
            for got, key in zip(r, ("hid", "enc", "cond")):
                assert relerr(got, G[f"attn_d_{mode}_{key}"]) < TOL
            r = fr.attn_forward(s, torch.cat([G["enc"], G["hid"]], 1), None, G["cond"], None, main, cond, mc)
            for got, key in zip(r, ("hid", "cond")):
                assert relerr(got, G[f"attn_s_{mode}_{key}"]) < TOL
    finally:
        for a in (d, s):
            if hasattr(a, "c_factor"):
                del a.c_factor


def test_attn_forward_nocond(G, tr):
    main, _ = _ropes(tr, G)
    with torch.no_grad():
        r = fr.attn_forward(tr.transformer_blocks[0].attn, G["hid"], G["enc"], None, None, main, None, {})
        assert relerr(r[0], G["attn_d_nocond_hid"]) < TOL and relerr(r[1], G["attn_d_nocond_enc"]) < TOL
        r = fr.attn_forward(tr.single_transformer_blocks[0].attn, torch.cat([G["enc"], G["hid"]], 1), None, None,
                            None, main, None, {})
        assert relerr(r, G["attn_s_nocond_hid"]) < TOL


@pytest.mark.parametrize("name,mc", [("default", {}), ("add_cond", {"add_cond_attn": True})])
def test_block_forward(G, tr, name, mc):
    main, cond = _ropes(tr, G)
    with torch.no_grad():
        e, h, c = fr.block_forward(tr.transformer_blocks[1], G["hid"], G["enc"], G["cond"], G["temb"], G["ctemb"],
                                   cond, main, mc)
    assert relerr(e, G[f"block_{name}_enc"]) < TOL
    assert relerr(h, G[f"block_{name}_hid"]) < TOL
    assert relerr(c, G[f"block_{name}_cond"]) < TOL


def test_block_forward_nocond(G, tr):
    main, _ = _ropes(tr, G)
    with torch.no_grad():
        e, h, c = fr.block_forward(tr.transformer_blocks[1], G["hid"], G["enc"], None, G["temb"], None, None, main, {})
    assert c is None and relerr(e, G["block_nocond_enc"]) < TOL and relerr(h, G["block_nocond_hid"]) < TOL


def test_single_block_forward(G, tr):
    main, cond = _ropes(tr, G)
    hs = torch.cat([G["enc"], G["hid"]], 1)
    with torch.no_grad():
        h, c = fr.single_block_forward(tr.single_transformer_blocks[1], hs, G["temb"], main, G["cond"], G["ctemb"], cond, {})
        h2 = fr.single_block_forward(tr.single_transformer_blocks[1], hs, G["temb"], main)
    assert relerr(h, G["single_hid"]) < TOL and relerr(c, G["single_cond"]) < TOL
    assert relerr(h2, G["single_nocond_hid"]) < TOL


@pytest.mark.parametrize("name,factor", [("lora_off", 0.0), ("lora_half", 0.5)])
def test_blocks_under_the_reference_lora_switches(G, tr, name, factor):
    """goldens made by the reference's enable_lora(modules, False) / set_lora_scale(modules, 0.5) (lora_controller.py:5-75) around
    its block functions; the restatement's blocks with every adapter's scaling multiplied by the same factor must agree."""
    from oracle import flux_modules as fm
    main, cond = _ropes(tr, G)
    dblk, sblk = tr.transformer_blocks[1], tr.single_transformer_blocks[1]
    layers = [m for b in (dblk, sblk) for m in b.modules() if isinstance(m, fm.LoraLinear)]
    saved = [dict(m.scaling) for m in layers]
    try:
        for m in layers:
            for a in m.active_adapters:
                m.scaling[a] *= factor
        hs = torch.cat([G["enc"], G["hid"]], 1)
        with torch.no_grad():
            e, h, c = fr.block_forward(dblk, G["hid"], G["enc"], G["cond"], G["temb"], G["ctemb"], cond, main, {})
            h1, c1 = fr.single_block_forward(sblk, hs, G["temb"], main, G["cond"], G["ctemb"], cond, {})
    finally:
        for m, sc in zip(layers, saved):
            m.scaling.update(sc)
    for got, key in ((e, f"block_{name}_enc"), (h, f"block_{name}_hid"), (c, f"block_{name}_cond"), (h1, f"single_{name}_hid"), (c1, f"single_{name}_cond")):
        assert relerr(got, G[key]) < TOL, key


def _fwd(tr, G, cond=True, c_t=0, guidance=True):
    with torch.no_grad():
        return fr.tranformer_forward(tr, G["in_cond"] if cond else None, G["in_cond_ids"] if cond else None, None, {},
                                     c_t, hidden_states=G["in_latents"], encoder_hidden_states=G["in_enc"],
                                     pooled_projections=G["in_pooled"], timestep=G["in_timestep"],
                                     img_ids=G["in_img_ids"], txt_ids=G["in_txt_ids"],
                                     guidance=G["in_guidance"] if guidance else None)[0]


def test_transformer_forward(G, tr):
    assert relerr(_fwd(tr, G), G["fwd_cond"]) < TOL
    assert relerr(_fwd(tr, G, cond=False), G["fwd_nocond"]) < TOL
    assert relerr(_fwd(tr, G, c_t=0.25), G["fwd_cond_ct025"]) < TOL


def test_transformer_forward_no_guidance(G):
    tr2 = tiny_transformer(seed=3, guidance_embeds=False)
    assert relerr(_fwd(tr2, G, guidance=False), G["fwd_noguidance_seed3"]) < TOL


# ------------------------------------------------------------------ generate() (reference src/flux/generate.py:72-394)
def _brain():
    from oracle import cs3 as ocs3
    torch.manual_seed(0)
    return ocs3.CS3DGF(seed=0).eval()


@pytest.mark.parametrize("case", [c[0] for c in __import__("oracle.ducks", fromlist=["x"]).generate_cases()])
def test_generate_side_logic_against_the_real_generate(case):
    """`oracle.flux_ref.denoise_loop` + `oracle.cs3.CS3DGF.brain_embeds` (what the GPU tests check the product's generate()
    against) vs goldens made by the reference's REAL generate() through a duck-typed pipeline / model: sigma schedule, Euler
    stepping, the brain branch with the literal replacement rule, DUAN fusion, condition_scale -> c_factor."""
    from oracle import ducks
    G = load("generate_tiny.npz")
    name, fuse_flag, use, cscale = next(c for c in ducks.generate_cases() if c[0] == case)
    tr, brain = ducks.generate_transformer(), _brain()
    sig = {k: (G["in_" + k].unsqueeze(0) if k in use else None) for k in ("eeg", "fnirs", "ppg", "motion")}
    with torch.no_grad():
        pe, pooled = brain.brain_embeds(G["in_pe"], G["in_pooled"], sig["eeg"], sig["fnirs"], sig["ppg"], sig["motion"],
                                        fuse_flag=fuse_flag, per_stream=False)
        if cscale != 1.0:
            for n, m in tr.named_modules():
                if n.endswith(".attn"):
                    m.c_factor = torch.ones(1, 1) * cscale
        sch = fm.FlowMatchEulerDiscreteScheduler()
        got = fr.denoise_loop(tr, sch, G["in_lat"], pe, pooled, torch.zeros(512, 3), fm.prepare_latent_image_ids(4, 4),
                              G["cond_tokens"], G["cond_ids"], num_inference_steps=4)
    assert relerr(got, G[f"gen_{case}"]) < 5e-6
    assert torch.equal(sch.timesteps, G["sched_timesteps"]) and torch.equal(sch.sigmas, G["sched_sigmas"])
