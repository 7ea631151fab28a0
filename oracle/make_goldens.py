#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference functions.

Runs ONLY in the build container (needs /root/reference).  Usage:

    python3 -B oracle/make_goldens.py

The reference's Python files never leave this container; only the input/output vectors
written here do.  Third-party modules the reference imports but that are not installed
(diffusers, peft, lightning, prodigyopt, s4torch, cv2) are replaced in sys.modules by
thin namespaces whose members are the `oracle/flux_modules.py` / `oracle/s4.py`
restatements, so what these goldens pin is the reference's OWN arithmetic
(src/flux/block.py, src/flux/transformer.py, src/train/model.py DUAN/FPP/fuse_*/encoders).
All weights are re-derivable from seeds (oracle.flux_modules.init_synthetic_), so the
fixtures hold only inputs, outputs and the config needed to rebuild the modules.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("LX_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import flux_modules as fm  # noqa: E402
from oracle import s4 as os4  # noqa: E402


def _ns(name, **members):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__dict__.update(members)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)      # transformers probes find_spec() of optional packages
    sys.modules[name] = m
    return m


def install_stubs():
    import logging
    log = logging.getLogger("lxref")

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _ns("diffusers")
    _ns("diffusers.models")
    _ns("diffusers.models.attention_processor", Attention=fm.Attention, F=F)
    _ns("diffusers.models.embeddings", apply_rotary_emb=fm.apply_rotary_emb)
    _ns("diffusers.models.transformers")
    _ns("diffusers.models.transformers.transformer_flux", FluxTransformer2DModel=fm.FluxTransformer2DModel,
        Transformer2DModelOutput=_Dummy, USE_PEFT_BACKEND=False, scale_lora_layers=lambda *a, **k: None,
        unscale_lora_layers=lambda *a, **k: None, logger=log)
    _ns("diffusers.pipelines", FluxPipeline=_Dummy)
    _ns("diffusers.pipelines.flux")
    _ns("diffusers.pipelines.flux.pipeline_flux", FluxPipelineOutput=_Dummy, calculate_shift=fm.calculate_shift,
        retrieve_timesteps=fm.retrieve_timesteps, np=np, logger=log)
    _ns("diffusers.utils", logging=logging)
    _ns("peft", LoraConfig=_Dummy, get_peft_model_state_dict=lambda *a, **k: {})
    _ns("peft.tuners")
    _ns("peft.tuners.tuners_utils", BaseTunerLayer=fm.BaseTunerLayer)
    _ns("lightning", LightningModule=nn.Module, Callback=object)
    _ns("prodigyopt")
    _ns("cv2")

    class S4ModelAdapter(os4.S4Model):
        def __init__(self, d_input, d_model, d_output, n_blocks, n, l_max):
            super().__init__(d_input, d_model, d_output, n_blocks, n, l_max, torch.Generator().manual_seed(1234))

    _ns("s4torch", S4Model=S4ModelAdapter)
    pkg = types.ModuleType("refsrc")
    pkg.__path__ = [os.path.join(REF, "src")]
    sys.modules["refsrc"] = pkg


def t2n(x):
    return x.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
TINY = dict(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=64,
            pooled_dim=32, guidance_embeds=True, lora=True)


def tiny_transformer(seed=0, **over):
    cfg = dict(TINY)
    cfg.update(over)
    tr = fm.FluxTransformer2DModel(**cfg)
    fm.init_synthetic_(tr, seed=seed, std=0.05, bias_std=0.02, norm_jitter=0.1)
    return tr.eval()


def tiny_inputs(seed=1, B=2, T=16, hw=4, cw=4):
    g = torch.Generator().manual_seed(seed)
    N, C = hw * hw, cw * cw
    lat = torch.randn(B, N, 64, generator=g)
    cond = torch.randn(B, C, 64, generator=g)
    enc = torch.randn(B, T, 64, generator=g) * 0.5
    pooled = torch.randn(B, 32, generator=g)
    txt_ids = torch.zeros(T, 3)
    img_ids = fm.prepare_latent_image_ids(hw, hw)
    cond_ids = fm.prepare_latent_image_ids(cw, cw)
    cond_ids[:, 2] += -cw  # position_delta = [0, -w/16] (inference.py:350-351, condition.py:126-130)
    return dict(latents=lat, cond=cond, enc=enc, pooled=pooled, txt_ids=txt_ids, img_ids=img_ids,
                cond_ids=cond_ids, timestep=torch.tensor([0.7, 0.35])[:B], guidance=torch.full((B,), 3.5))


def gold_flux():
    from refsrc.flux import block as rb
    from refsrc.flux import transformer as rt
    tr = tiny_transformer()
    x = tiny_inputs()
    D = tr.inner_dim
    g = torch.Generator().manual_seed(7)
    B, T, N, C = 2, 16, 16, 16
    hid = torch.randn(B, N, D, generator=g)
    enc = torch.randn(B, T, D, generator=g)
    cond = torch.randn(B, C, D, generator=g)
    temb = torch.randn(B, D, generator=g)
    ctemb = torch.randn(B, D, generator=g)
    rope_main = tr.pos_embed(torch.cat([x["txt_ids"], x["img_ids"]], 0))
    rope_cond = tr.pos_embed(x["cond_ids"])
    out = dict(hid=t2n(hid), enc=t2n(enc), cond=t2n(cond), temb=t2n(temb), ctemb=t2n(ctemb))
    with torch.no_grad():
        # attn_forward, every mask mode, double + single flavour
        modes = {"default": ({}, None), "no_union": ({"union_cond_attn": False}, None),
                 "independent": ({"independent_condition": True}, None),
                 "cfactor_half": ({}, 0.5), "cfactor_two": ({}, 2.0), "latent_lora": ({"latent_lora": True}, None)}
        dattn, sattn = tr.transformer_blocks[0].attn, tr.single_transformer_blocks[0].attn
        for name, (mc, cf) in modes.items():
            for a in (dattn, sattn):
                if cf is not None:
                    a.c_factor = torch.ones(1, 1) * cf
            r = rb.attn_forward(dattn, hid, enc, cond, None, rope_main, rope_cond, mc)
            out[f"attn_d_{name}_hid"], out[f"attn_d_{name}_enc"], out[f"attn_d_{name}_cond"] = map(t2n, r)
            hs = torch.cat([enc, hid], 1)
            r = rb.attn_forward(sattn, hs, None, cond, None, rope_main, rope_cond, mc)
            out[f"attn_s_{name}_hid"], out[f"attn_s_{name}_cond"] = map(t2n, r)
            for a in (dattn, sattn):
                if hasattr(a, "c_factor"):
                    del a.c_factor
        r = rb.attn_forward(dattn, hid, enc, None, None, rope_main, None, {})
        out["attn_d_nocond_hid"], out["attn_d_nocond_enc"] = map(t2n, r)
        out["attn_s_nocond_hid"] = t2n(rb.attn_forward(sattn, torch.cat([enc, hid], 1), None, None, None, rope_main, None, {}))
        # blocks
        for name, mc in {"default": {}, "add_cond": {"add_cond_attn": True}}.items():
            e, h, c = rb.block_forward(tr.transformer_blocks[1], hid, enc, cond, temb, ctemb, rope_cond, rope_main, mc)
            out[f"block_{name}_enc"], out[f"block_{name}_hid"], out[f"block_{name}_cond"] = t2n(e), t2n(h), t2n(c)
        e, h, c = rb.block_forward(tr.transformer_blocks[1], hid, enc, None, temb, None, None, rope_main, {})
        assert c is None
        out["block_nocond_enc"], out["block_nocond_hid"] = t2n(e), t2n(h)
        hs = torch.cat([enc, hid], 1)
        h, c = rb.single_block_forward(tr.single_transformer_blocks[1], hs, temb, rope_main, cond, ctemb, rope_cond, {})
        out["single_hid"], out["single_cond"] = t2n(h), t2n(c)
        out["single_nocond_hid"] = t2n(rb.single_block_forward(tr.single_transformer_blocks[1], hs, temb, rope_main))
        # the reference's LoRA switches (src/flux/lora_controller.py:5-75) around its own block functions: adapters off on every
        # stream, and every adapter term halved
        from refsrc.flux import lora_controller as rl
        dblk, sblk = tr.transformer_blocks[1], tr.single_transformer_blocks[1]
        for name, ctx in {"lora_off": lambda m: rl.enable_lora(m, False), "lora_half": lambda m: rl.set_lora_scale(m, 0.5)}.items():
            with ctx(list(dblk.modules())):
                e, h, c = rb.block_forward(dblk, hid, enc, cond, temb, ctemb, rope_cond, rope_main, {})
            out[f"block_{name}_enc"], out[f"block_{name}_hid"], out[f"block_{name}_cond"] = t2n(e), t2n(h), t2n(c)
            with ctx(list(sblk.modules())):
                h, c = rb.single_block_forward(sblk, hs, temb, rope_main, cond, ctemb, rope_cond, {})
            out[f"single_{name}_hid"], out[f"single_{name}_cond"] = t2n(h), t2n(c)
        e, h, c = rb.block_forward(dblk, hid, enc, cond, temb, ctemb, rope_cond, rope_main, {})
        assert np.array_equal(t2n(h), out["block_default_hid"]) and np.array_equal(t2n(c), out["block_default_cond"])   # scales restored on exit
        # full forward
        for name, kw in {"cond": dict(c=True, g=True), "nocond": dict(c=False, g=True)}.items():
            r = rt.tranformer_forward(tr, x["cond"] if kw["c"] else None, x["cond_ids"] if kw["c"] else None,
                                      None, {}, hidden_states=x["latents"], encoder_hidden_states=x["enc"],
                                      pooled_projections=x["pooled"], timestep=x["timestep"], img_ids=x["img_ids"],
                                      txt_ids=x["txt_ids"], guidance=x["guidance"], return_dict=False)
            out[f"fwd_{name}"] = t2n(r[0])
        r = rt.tranformer_forward(tr, x["cond"], x["cond_ids"], None, {}, c_t=0.25, hidden_states=x["latents"],
                                  encoder_hidden_states=x["enc"], pooled_projections=x["pooled"],
                                  timestep=x["timestep"], img_ids=x["img_ids"], txt_ids=x["txt_ids"],
                                  guidance=x["guidance"], return_dict=False)
        out["fwd_cond_ct025"] = t2n(r[0])
        tr2 = tiny_transformer(seed=3, guidance_embeds=False)
        r = rt.tranformer_forward(tr2, x["cond"], x["cond_ids"], None, {}, hidden_states=x["latents"],
                                  encoder_hidden_states=x["enc"], pooled_projections=x["pooled"],
                                  timestep=x["timestep"], img_ids=x["img_ids"], txt_ids=x["txt_ids"],
                                  guidance=None, return_dict=False)
        out["fwd_noguidance_seed3"] = t2n(r[0])
    for k, v in x.items():
        out["in_" + k] = t2n(v)
    np.savez_compressed(os.path.join(OUT, "flux_tiny.npz"), **{k: v.astype(np.float32) for k, v in out.items()})
    print("flux_tiny.npz", len(out), "arrays")


def gold_cs3():
    from refsrc.train import model as rm
    out = {}
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        # DUAN
        for name, (C, L, hid) in {"c16": (16, 48, 8), "c1": (1, 768, 128), "c512": (512, 40, 128)}.items():
            torch.manual_seed(100 + C)
            d = rm.DUAN(C, hidden_dim=hid)
            x, c = torch.randn(2, C, L, generator=g), torch.randn(2, C, L, generator=g)
            out[f"duan_{name}_x"], out[f"duan_{name}_c"], out[f"duan_{name}_y"] = t2n(x), t2n(c), t2n(d(x, c))
            out[f"duan_{name}_seed"] = np.array([100 + C, hid])  # weights: torch.manual_seed(seed); DUAN(C, hid)
        # FPP + adaptive pools on the true lengths / size sets
        for name, (ch, L, sizes) in {"eeg": (4, 4096, [128, 256, 512, 1024, 2048]), "ppg": (4, 256, [64, 128, 256]),
                                     "fnirs": (6, 512, [128, 256, 448]), "motion": (6, 128, [32, 64, 124])}.items():
            x = torch.randn(2, ch, L, generator=g)
            out[f"fpp_{name}_x"], out[f"fpp_{name}_y"] = t2n(x), t2n(rm.FeaturePyramidPooling(sizes)(x))
        # spatial_pyramid_pooling pad / truncate / equal
        x = torch.randn(2, 3, 50, generator=g)
        out["spp_x"] = t2n(x)
        for n, o in {"pad": 64, "trunc": 32, "same": 50}.items():
            out[f"spp_{n}"] = t2n(rm.OminiModel.spatial_pyramid_pooling(None, x, o))
        # fuse_eeg / fuse_fnirs through the unbound reference methods
        torch.manual_seed(5)
        ns = types.SimpleNamespace(duan_norm1=rm.DUAN(512), fusion1=nn.Sequential(nn.Linear(1024, 512)),
                                   duan_norm2=rm.DUAN(1), fusion2=nn.Sequential(nn.Linear(1536, 768)))
        e, p = torch.randn(1, 512, 64, generator=g), torch.randn(1, 512, 64, generator=g)
        out["fuse_eeg_e"], out["fuse_eeg_p"] = t2n(e), t2n(p)
        out["fuse_eeg_y"] = t2n(rm.OminiModel.fuse_eeg(ns, e, p))
        f, m = torch.randn(2, 768, generator=g), torch.randn(2, 768, generator=g)
        out["fuse_fnirs_f"], out["fuse_fnirs_m"] = t2n(f), t2n(m)
        out["fuse_fnirs_y"] = t2n(rm.OminiModel.fuse_fnirs(ns, f, m))
        out["fuse_seed"] = np.array([5])  # torch.manual_seed(5); DUAN(512), Linear(1024,512), DUAN(1), Linear(1536,768)
    np.savez_compressed(os.path.join(OUT, "cs3_dgf.npz"), **{k: v.astype(np.float32) for k, v in out.items()})
    print("cs3_dgf.npz", len(out), "arrays")


def gold_encoders():
    """Reference encoder classes with oracle S4 standing in for s4torch (pins the wrapper:
    permutes, pools, FPP concat order, MLP head).  Small ones only (weights re-derived by seed)."""
    from refsrc.train import model as rm
    out = {}
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for name, cls, shape in (("ppg", rm.PPGEncoder, (2, 4, 256)), ("fnirs", rm.FNIRSEncoder, (2, 6, 512)),
                                 ("motion", rm.MotionEncoder, (2, 6, 128))):
            torch.manual_seed(77)
            enc = cls(device="cpu", dtype=torch.float32).eval()
            x = torch.randn(*shape, generator=g)
            y = enc(x)
            out[f"enc_{name}_x"] = t2n(x)
            if y.dim() == 3:   # [B,512,4096] is 8 MB: keep a strided sample + full checksum
                out[f"enc_{name}_y_sample"] = t2n(y[:, ::37, ::53])
                out[f"enc_{name}_y_sum"] = np.array([float(y.double().sum()), float(y.double().abs().sum())])
            else:
                out[f"enc_{name}_y"] = t2n(y)
            out[f"enc_{name}_seed"] = np.array([77, 1234])
    np.savez_compressed(os.path.join(OUT, "cs3_encoders.npz"), **{k: np.asarray(v, dtype=np.float64 if k.endswith("_sum") else np.float32) for k, v in out.items()})
    print("cs3_encoders.npz", len(out), "arrays")


def gold_eeg_encoder():
    """The reference EEGEncoder class (model.py:16-134; oracle S4 standing in for s4torch) -- the one encoder BASELINE
    configs[1] runs. 33.5 M parameters re-derived by seed; the [1,512,4096] output kept as a strided sample + checksums."""
    from refsrc.train import model as rm
    out = {}
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        torch.manual_seed(78)
        enc = rm.EEGEncoder(device="cpu", dtype=torch.float32).eval()
        x = torch.randn(2, 4, 4096, generator=g)
        y = enc(x)
        assert y.shape == (2, 512, 4096)
        out["enc_eeg_x"] = t2n(x)
        out["enc_eeg_y_sample"] = t2n(y[:, ::37, ::53])
        out["enc_eeg_y_row0"] = t2n(y[:, 0, :256])
        out["enc_eeg_y_sum"] = np.array([float(y.double().sum()), float(y.double().abs().sum())])
        out["enc_eeg_seed"] = np.array([78, 1234])      # torch.manual_seed(78); both S4Model instances draw from Generator(1234)
    np.savez_compressed(os.path.join(OUT, "cs3_eeg_encoder.npz"),
                        **{k: np.asarray(v, dtype=np.float64 if k.endswith("_sum") else np.float32) for k, v in out.items()})
    print("cs3_eeg_encoder.npz", len(out), "arrays")


def gold_condition():
    """Condition.encode (condition.py:106-138) + encode_images (pipeline_tools.py:7-30) through a duck-typed pipeline:
    tokens, position ids (default subject delta, explicit delta, position_scale affine) and type ids."""
    from refsrc.flux import condition as rc
    from oracle import ducks
    out = {}
    pipe = ducks.DuckFluxPipeline(None)
    cases = {"subject_default": dict(condition_type="subject", wh=(64, 48)),
             "subject_delta": dict(condition_type="subject", wh=(64, 64), position_delta=[0, -32]),
             "subject_delta_rc": dict(condition_type="subject", wh=(32, 64), position_delta=[3, -5]),
             "subject_scale2": dict(condition_type="subject", wh=(64, 64), position_scale=2.0),
             "fill_scale_half": dict(condition_type="fill", wh=(96, 64), position_delta=[0, 0], position_scale=0.5),
             "cartoon_nodelta": dict(condition_type="cartoon", wh=(64, 64))}
    for name, kw in cases.items():
        kw = dict(kw)
        w, h = kw.pop("wh")
        c = rc.Condition(condition=ducks.DuckImage(w, h, seed=len(name)), **kw)
        tokens, ids, type_id = c.encode(pipe)
        out[f"{name}_tokens"], out[f"{name}_ids"], out[f"{name}_type"] = t2n(tokens), t2n(ids), t2n(type_id)
        out[f"{name}_cfg"] = np.array([w, h, len(name), kw.get("position_scale", 1.0)] + list(kw.get("position_delta") or [np.nan, np.nan]))
    try:
        rc.Condition(condition_type="eeg+fnirs", condition=ducks.DuckImage(64, 64)).encode(pipe)
        raise AssertionError("expected NotImplementedError")
    except NotImplementedError:
        pass
    np.savez_compressed(os.path.join(OUT, "condition_ids.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in out.items()})
    print("condition_ids.npz", len(out), "arrays")


def gold_generate():
    """The REAL generate() (src/flux/generate.py:72-394) on the tiny 2+2-block config, 4 steps, latents in / latents out,
    driven with a duck-typed pipeline and a duck-typed OminiModel: sigma schedule, Euler stepping, the brain branch
    (unsqueeze(0) -> spatial_pyramid_pooling -> encoders -> fuse_eeg / fuse_fnirs -> DUAN fusion or replacement), the
    condition_scale -> c_factor hook and Condition.encode. Documented delta Q1: the reference hands the encoders
    `signal.flatten(1)`, which its own encoder classes cannot consume ([B,C,L] permutes): the duck encoders un-flatten, i.e.
    the golden pins generate() AS IF the encoders received [B,C,L] -- what the product (and OminiModel.step) does."""
    from functools import partial
    from refsrc.flux import condition as rc
    from refsrc.flux import generate as rg
    from refsrc.train import model as rm
    from oracle import cs3 as ocs3
    from oracle import ducks
    tr = ducks.generate_transformer()
    torch.manual_seed(0)
    brain = ocs3.CS3DGF(seed=0).eval()        # the same construction the tests use: weights re-derivable from the seeds
    real = {}
    for n, C in (("duan_norm1", 512), ("duan_norm2", 1), ("duan_norm_prompt", 512), ("duan_norm_pooled", 1)):
        real[n] = rm.DUAN(C).eval()            # the reference's own class, carrying the seeded weights
        real[n].load_state_dict(getattr(brain, n).state_dict())
    ns = types.SimpleNamespace(fusion1=brain.fusion1, fusion2=brain.fusion2, **real)
    unflat = lambda enc, ch: (lambda x: enc(x.view(x.shape[0], ch, -1)))
    model = types.SimpleNamespace(
        eeg_projection=unflat(brain.eeg_projection, 4), ppg_projection=unflat(brain.ppg_projection, 4),
        fnirs_projection=unflat(brain.fnirs_projection, 6), motion_projection=unflat(brain.motion_projection, 6),
        fuse_eeg=partial(rm.OminiModel.fuse_eeg, ns), fuse_fnirs=partial(rm.OminiModel.fuse_fnirs, ns),
        spatial_pyramid_pooling=partial(rm.OminiModel.spatial_pyramid_pooling, None),
        eeg_fixed_length=4096, fnirs_fixed_length=512, ppg_fixed_length=256, motion_fixed_length=128, **real)
    x = ducks.generate_inputs()
    hw = 4
    out = {"in_" + k: t2n(v) for k, v in x.items()}
    with torch.no_grad():
        for name, fuse_flag, use, cscale in ducks.generate_cases():
            pipe = ducks.DuckFluxPipeline(tr)
            cond = rc.Condition(condition_type="subject", condition=ducks.DuckImage(hw * 16, hw * 16, seed=5))
            sig = {k: (x[k] if k in use else None) for k in ("eeg", "fnirs", "ppg", "motion")}
            r = rg.generate(model, pipe, conditions=[cond], height=hw * 16, width=hw * 16, num_inference_steps=4,
                            latents=x["lat"].clone(), prompt_embeds=x["pe"], pooled_prompt_embeds=x["pooled"], output_type="latent",
                            model_config={}, default_lora=True, condition_scale=cscale, additional_condition1=sig["eeg"],
                            additional_condition2=sig["fnirs"], additional_condition3=sig["ppg"], additional_condition4=sig["motion"],
                            use_brain_condition=bool(use), fuse_flag=fuse_flag, return_dict=False)
            out[f"gen_{name}"] = t2n(r[0])
            assert not any(hasattr(m, "c_factor") for _, m in tr.named_modules())     # removed again on exit (generate.py:384-388)
        out["sched_timesteps"], out["sched_sigmas"] = t2n(pipe.scheduler.timesteps), t2n(pipe.scheduler.sigmas)
        # the condition tokens/ids generate() fed the transformer (from the duck VAE): handed to the product as Condition(latents=...)
        tokens, ids, _ = rc.Condition(condition_type="subject", condition=ducks.DuckImage(hw * 16, hw * 16, seed=5)).encode(pipe)
        out["cond_tokens"], out["cond_ids"] = t2n(tokens), t2n(ids)
    assert np.array_equal(out["gen_plain"], out["gen_eeg_only"])       # the literal rule ignores a lone EEG (generate.py:252-255)
    np.savez_compressed(os.path.join(OUT, "generate_tiny.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in out.items()})
    print("generate_tiny.npz", len(out), "arrays")


def _install_eval_stubs():
    """torchvision / clip are not installed: the reference's test.py imports them at module level. ToTensor / Compose / Resize /
    CenterCrop / Normalize are restated with torchvision's semantics (Resize: shorter edge -> size, longer edge int(size * long /
    short); CenterCrop: top = int(round((h - size) / 2.0))); `clip` is imported by test.py but never used."""
    from PIL import Image

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=2):
            self.size, self.interp = size, interpolation

        def __call__(self, img):
            w, h = img.size
            short, long = (w, h) if w <= h else (h, w)
            ns, nl = self.size, int(self.size * long / short)
            nw, nh = (ns, nl) if w <= h else (nl, ns)
            return img.resize((nw, nh), resample={2: Image.BILINEAR, 3: Image.BICUBIC}[self.interp])

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            w, h = img.size
            t, l = int(round((h - self.size) / 2.0)), int(round((w - self.size) / 2.0))
            return img.crop((l, t, l + self.size, t + self.size))

    class Normalize:
        def __init__(self, mean, std):
            self.m, self.s = torch.tensor(mean)[:, None, None], torch.tensor(std)[:, None, None]

        def __call__(self, x):
            return (x - self.m) / self.s

    from transformers import CLIPModel, CLIPProcessor  # noqa: F401  (resolve transformers' lazy imports before torchvision is stubbed)
    tv = _ns("torchvision")
    tt = _ns("torchvision.transforms.transforms", ToTensor=ToTensor, Compose=Compose, Resize=Resize, CenterCrop=CenterCrop, Normalize=Normalize)
    tr = _ns("torchvision.transforms", transforms=tt)
    tr.functional = _ns("torchvision.transforms.functional")
    tv.transforms = tr
    _ns("clip")
    return tt


def gold_evaluate():
    """The reference's evaluator (test.py:17-214 functions, :241-249 pairing, :321-336 result files) on a synthetic image set:
    L1 / L2, CLIP-I, CLIP-T (incl. the caption lookup) with a tiny seeded CLIP, DINO with a stand-in backbone, and main() with
    --metric l1,l2. The fixture holds the images, captions, model weights and the reference's outputs."""
    import contextlib
    import importlib.util
    import io
    import json
    import tempfile
    import types as _t
    from oracle import ducks
    tt = _install_eval_stubs()
    spec = importlib.util.spec_from_file_location("ref_test", os.path.join(REF, "test.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    rt.tqdm = lambda x: x
    gen, gt, caps = ducks.evaluator_images()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        gdir, tdir, cap = ducks.write_evaluator_dirs(d, gen, gt, caps)
        from transformers import CLIPModel, CLIPProcessor
        clip_dir = os.path.join(d, "clip")
        ducks.tiny_clip(clip_dir)
        model5, proc = CLIPModel.from_pretrained(clip_dir).eval(), CLIPProcessor.from_pretrained(clip_dir)

        class Clip4x:
            """The reference is written against transformers 4.x, where get_image_features / get_text_features return the
            projected feature TENSOR; the installed 5.x returns a model output whose .pooler_output is that tensor (checked:
            == visual_projection(vision_model(x).pooler_output)). Present the 4.x surface to the reference's functions."""
            def get_image_features(self, pixel_values):
                return model5.get_image_features(pixel_values).pooler_output

            def get_text_features(self, input_ids):
                return model5.get_text_features(input_ids).pooler_output

            def state_dict(self):
                return model5.state_dict()
        model = Clip4x()
        dino = ducks.tiny_dino()
        # the pairing rule of main() (test.py:241-249)
        pairs = []
        for name in sorted(os.listdir(gdir)):
            if name.endswith((".png", ".jpg")):
                g = os.path.join(tdir, name.replace("_0", "_1"))
                if os.path.exists(g):
                    pairs.append((os.path.join(gdir, name), g))
        args = _t.SimpleNamespace(device=torch.device("cpu"))
        names = [os.path.basename(p[0]) for p in pairs]
        for m in ("l1", "l2"):
            s, res = rt.eval_distance(pairs, m)
            out[f"{m}_mean"], out[f"{m}_per_image"] = s, [res[n][m] for n in names]
        s, res = rt.eval_clip_i(args, pairs, model, proc)
        out["clip_i_mean"], out["clip_i_per_image"] = s, [res[n]["clip_i"] for n in names]
        dproc = tt.Compose([tt.Resize(256, interpolation=3), tt.CenterCrop(224), tt.ToTensor(),
                            tt.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])          # test.py:293-298
        s, res = rt.eval_dino_i(args, pairs, dino, dproc, metric="dino")
        out["dino_mean"], out["dino_per_image"] = s, [res[n]["dino"] for n in names]
        g_t, t_t, res = rt.eval_clip_t(args, pairs, model, proc, caps)
        out["clip_t_gen"], out["clip_t_gt"], out["clip_t_per_image"] = g_t, t_t, [res[n]["clip-t"] for n in names]
        # main() end to end with the metrics that need no external weights: pairing + result files
        save = os.path.join(d, "res")
        argv = sys.argv
        sys.argv = ["test.py", "--device", "cpu", "--caption_path", cap, "--generated_path", gdir, "--gt_path", tdir, "--metric", "l1,l2", "--save_path", save]
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                rt.main()
        finally:
            sys.argv = argv
        files = sorted(os.listdir(save))
        metrics_txt = open(os.path.join(save, "evaluation_metrics.txt")).read()
        csv_txt = open(os.path.join(save, "per_image_metrics.csv")).read()
        sd = {k: t2n(v) for k, v in model.state_dict().items()}
        dsd = {k: t2n(v) for k, v in dino.state_dict().items()}
    arrays = {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}
    arrays["pair_names"] = np.array(names)
    arrays["result_files"] = np.array(files)
    arrays["metrics_txt"], arrays["csv_txt"] = np.array(metrics_txt), np.array(csv_txt)
    arrays["captions_json"] = np.array(json.dumps(caps))
    for n, a in gen.items():
        arrays["gen/" + n] = a
    for n, a in gt.items():
        arrays["gt/" + n] = a
    for k, v in sd.items():
        arrays["clip/" + k] = v
    for k, v in dsd.items():
        arrays["dino/" + k] = v
    np.savez_compressed(os.path.join(OUT, "evaluate.npz"), **arrays)
    print("evaluate.npz", len(arrays), "arrays;", {k: float(np.round(v, 6)) for k, v in out.items() if np.ndim(v) == 0})


def gold_inference():
    """The host logic of the reference's inference CLI (inference.py:63-176, 264-339): caption selection, brain-data lookup, what
    reaches generate() per image, output files, the static shard rule -- the reference's own functions with `generate` / `Condition`
    replaced by a recorder. The fixture is the recorded decisions (JSON)."""
    import importlib.util
    import json
    import tempfile
    import types as _t
    from oracle import ducks
    _ns("accelerate", init_empty_weights=None, infer_auto_device_map=None)
    saved = {k: sys.modules.get(k) for k in ("src", "src.flux", "src.flux.condition", "src.flux.generate", "src.train", "src.train.model")}
    rec = ducks.GenerateRecorder()
    try:
        # the reference imports `src.*`: hand it thin modules (its own src would pull the whole stack in; only three names are used)
        _ns("src"); _ns("src.flux"); _ns("src.train")
        _ns("src.flux.condition", Condition=lambda **kw: rec.condition(**kw))
        _ns("src.flux.generate", generate=rec.generate)
        _ns("src.train.model", OminiModel=object)
        spec = importlib.util.spec_from_file_location("ref_inference", os.path.join(REF, "inference.py"))
        ri = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ri)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    ri.tqdm = lambda x, **k: x
    images, caps, brain = ducks.inference_case()
    model = _t.SimpleNamespace(device=torch.device("cpu"), flux_pipe=object(), model_config={"union_cond_attn": True, "latent_lora": False})
    out = {}
    with tempfile.TemporaryDirectory() as d:
        idir, cap, pkl = ducks.write_inference_case(d, images, caps, brain)
        # (1) batch_inference: one process, every captioned image
        o1 = os.path.join(d, "out1")
        ri.batch_inference(model, idir, o1, caption_path=cap, condition_type="subject", target_size=256, position_delta=[0, -16], seed=7,
                           brain_data_path=pkl)
        out["batch"] = dict(calls=rec.calls, files=sorted(os.listdir(o1)))
        # (2) process_image_batch on 2 and 3 ranks: the static shard rule + per-rank outputs
        bd = ri.load_brain_data(pkl)
        captions = {}
        with open(cap) as f:
            for line in f:
                item = json.loads(line)
                name = os.path.basename(item.get("source_image", ""))
                captions[name] = item["speech2text"] if "speech2text" in item else item.get("instruction", "Edit this image")
        files = [f for f in captions if f.endswith((".png", ".jpg", ".jpeg"))]
        out["image_files"] = files
        for world in (2, 3):
            per_rank = []
            for rank in range(world):
                rec.calls = []
                o = os.path.join(d, f"out_w{world}_r{rank}")
                os.makedirs(o)
                ri.process_image_batch(rank, world, model, files, idir, o, captions, bd, "subject", [0, -16], 256, 11)
                per_rank.append(dict(calls=rec.calls, files=sorted(os.listdir(o))))
            out[f"world{world}"] = per_rank
        out["missing_brain_file"] = ri.load_brain_data(os.path.join(d, "nope.pkl"))
    with open(os.path.join(OUT, "inference_cli.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("inference_cli.json:", len(out["batch"]["calls"]), "batch calls;", [len(r["calls"]) for r in out["world2"]], [len(r["calls"]) for r in out["world3"]])


if __name__ == "__main__":
    assert os.path.isdir(REF), f"{REF} not found: goldens can only be regenerated in the build container"
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    for fn in (gold_flux, gold_cs3, gold_encoders, gold_eeg_encoder, gold_condition, gold_generate, gold_evaluate, gold_inference):
        if not only or fn.__name__ in only:
            fn()
