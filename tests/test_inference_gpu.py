"""The reference-style entry points end to end on MI355X: `OminiModel(flux_pipe_id=<local dir>, ...)` -> `load_state_dict` ->
`inference_single_image(PIL image, prompt string)` -> PIL image (reference inference.py:24-121, src/train/model.py:376-477),
through the real chain  prompt -> CLIP/T5 (transformers on ROCm) -> Condition.encode (HIP VAE encode) -> 4-step denoise (HIP DiT) ->
HIP VAE decode -> PIL;  and `inference.py --synthetic` in process."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_modules as fm  # noqa: E402
from oracle import flux_ref as fr  # noqa: E402
from tests import tiny_ckpt  # noqa: E402
from tests.helpers import relerr  # noqa: E402


def _img(w, h, seed=0):
    from PIL import Image
    return Image.fromarray((np.random.default_rng(seed).random((h, w, 3)) * 255).astype("uint8"))


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = str(tmp_path_factory.mktemp("flux_tiny"))
    tr, vae = tiny_ckpt.build_flux_dir(d)
    sd, brain = tiny_ckpt.loongx_state_dict(tr)
    return d, tr, vae, sd, brain


def test_reference_style_construction_and_checkpoint_load(ckpt):
    from src.train.model import OminiModel
    d, tr, vae, sd, brain = ckpt
    model = OminiModel(flux_pipe_id=d, lora_config={"r": 4, "lora_alpha": 4}, device="cuda", dtype=torch.bfloat16,
                       model_config={"union_cond_attn": True})
    assert model.flux_pipe.vae is not None and model.flux_pipe.text_encoder is not None
    assert not model.transformer.engine.w.lora                       # the pipeline directory holds the base model only
    with pytest.raises(KeyError):
        model.load_state_dict(dict(sd, bogus=torch.zeros(1)))
    assert model.load_state_dict(sd) is model
    assert model.transformer.engine.w.lora and model.flux_pipe.transformer is model.transformer
    assert model.to("cuda") is model and model.eval() is model and model.device.type == "cuda"
    assert model.eeg_fixed_length == 4096 and model.model_config == {"union_cond_attn": True}
    # dtype float32 (the shipped config) selects the precise mode
    m32 = OminiModel(flux_pipe_id=None, device="cuda", dtype=torch.float32, flux_config=model.transformer.engine.cfg).load_state_dict(sd)
    assert m32.transformer.engine.precise_default and m32.transformer.engine.w.precise_ready


def test_inference_single_image_pil_to_pil_matches_oracle_chain(ckpt, monkeypatch, tmp_path):
    import inference as inf
    from loongx_amd import vae as lxvae
    from oracle import vae as ovae
    from src.train.model import OminiModel
    d, tr, vae, sd, brain = ckpt
    # the VAE posterior sample draws from the device RNG: take the mean on both sides so the chain is comparable
    monkeypatch.setattr(lxvae.DiagonalGaussianDistribution, "sample", lambda self, generator=None, noise=None: self.mean)
    model = OminiModel(flux_pipe_id=d, lora_config={"r": 4, "lora_alpha": 4}, device="cuda", dtype=torch.bfloat16, model_config={})
    model.load_state_dict(sd)
    size, steps = 64, 4
    cimg, prompt = _img(size, size, 3), "make the sky red"
    lat0 = torch.randn(1, 16, 64, generator=torch.Generator().manual_seed(9))
    out = inf.inference_single_image(model, cimg, prompt, condition_type="subject", position_delta=[0, -4], target_size=size, seed=1,
                                     latents=lat0.cuda(), num_inference_steps=steps)
    assert out.size == (size, size) and out.mode == "RGB"
    again = inf.inference_single_image(model, cimg, prompt, condition_type="subject", position_delta=[0, -4], target_size=size, seed=1,
                                       latents=lat0.cuda(), num_inference_steps=steps)
    assert np.array_equal(np.asarray(out), np.asarray(again))
    # ---- the same chain on the oracle side (fp32, CPU) ----
    te = model.flux_pipe.text_encoder
    with torch.no_grad():
        pe = te.t5_sequence([prompt]).float().cpu()
        pooled = te.clip_pooled([prompt]).float().cpu()
        x = model.flux_pipe.image_processor.preprocess(cimg)
        z = (vae.encode(x).latent_dist.mean - vae.config.shift_factor) * vae.config.scaling_factor
        tokens = fm.pack_latents(z)
        ids = fm.prepare_latent_image_ids(4, 4)
        cids = ids.clone()
        cids[:, 2] -= 4
        final = fr.denoise_loop(tr, fm.FlowMatchEulerDiscreteScheduler(), lat0, pe, pooled, torch.zeros(512, 3), ids, tokens, cids,
                                num_inference_steps=steps)
        img = vae.decode(fm.unpack_latents(final, size, size) / vae.config.scaling_factor + vae.config.shift_factor, return_dict=False)[0]
    want = (img / 2 + 0.5).clamp(0, 1)[0].permute(1, 2, 0).numpy()
    got = np.asarray(out).astype(np.float32) / 255.0
    assert np.abs(got - want).mean() < 0.02, float(np.abs(got - want).mean())          # 8-bit pixels, bf16 VAE + DiT vs fp32
    # the evaluator (reference test.py; loongx_amd/evaluate.py, pinned to the reference by tests/golden/evaluate.npz) scores the
    # product's image against the oracle chain's image: the `_0` / `_1` pairing, L1 / L2, CLIP-I with a tiny CLIP on the GPU
    import types
    from PIL import Image
    from transformers import CLIPModel, CLIPProcessor
    from loongx_amd.evaluate import collect_pairs, eval_clip_i, eval_distance
    from oracle import ducks
    gdir, tdir = tmp_path / "generated", tmp_path / "gt"
    gdir.mkdir(); tdir.mkdir()
    out.save(gdir / "edit_0.png")
    Image.fromarray((want * 255.0).round().astype("uint8")).save(tdir / "edit_1.png")
    pairs = collect_pairs(str(gdir), str(tdir))
    assert len(pairs) == 1
    l1, _ = eval_distance(pairs, "l1")
    l2, _ = eval_distance(pairs, "l2")
    assert l1 < 0.02 and l2 < 0.02 ** 2 * 4, (l1, l2)
    ducks.tiny_clip(str(tmp_path / "clip"))
    clip = CLIPModel.from_pretrained(str(tmp_path / "clip")).eval().cuda()
    ci, _ = eval_clip_i(types.SimpleNamespace(device=torch.device("cuda")), pairs, clip, CLIPProcessor.from_pretrained(str(tmp_path / "clip")))
    assert ci > 0.99, ci
    # prompt strings reach the model: a different prompt gives a different image
    other = inf.inference_single_image(model, cimg, "a blue cat", condition_type="subject", position_delta=[0, -4], target_size=size,
                                       seed=1, latents=lat0.cuda(), num_inference_steps=steps)
    assert not np.array_equal(np.asarray(out), np.asarray(other))


def test_inference_cli_synthetic_in_process(tmp_path, monkeypatch):
    """`python inference.py --synthetic --num_images 2` (one rank), and the written latents against a direct generate() call."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import inference as inf
    from loongx_amd.flux.weights import FluxConfig
    from src.train.model import OminiModel
    small = FluxConfig(num_layers=1, num_single_layers=1)
    made = {}

    def synthetic(cls=None, flux_config=None, model_config=None, device="cuda", seed=0, dtype=torch.bfloat16):
        made["m"] = OminiModel.synthetic(small, model_config, device, seed, dtype)
        return made["m"]
    monkeypatch.setattr(inf, "load_model", lambda ckpt, config=None, device=None: synthetic(model_config=(config or {}).get("model", {}), device=device))
    out = str(tmp_path / "out")
    inf.main(["--synthetic", "--num_images", "2", "--num_gpus", "1", "--output_dir", out, "--target_size", "256", "--position_delta_y", "-16"])
    files = sorted(os.listdir(out))
    assert files == ["synthetic_00000.latent.pt", "synthetic_00001.latent.pt"]
    lat = torch.load(os.path.join(out, files[1]))
    item = inf.synthetic_item(1, 256, torch.device("cuda", 0), 42)
    want = inf.inference_single(made["m"], item, "subject", [0, -16], 256)
    assert lat.shape == (256, 64) and torch.equal(lat, want.cpu())
    # round 6: `--operands fp16 --f16-overflow fallback` reaches the engine through the config's model section, every saved image passed the
    # synchronous saturation check, and the fp16 latents are the bf16 ones to the two formats' rounding difference
    out16 = str(tmp_path / "out16")
    inf.main(["--synthetic", "--num_images", "2", "--num_gpus", "1", "--output_dir", out16, "--target_size", "256", "--position_delta_y", "-16",
              "--operands", "fp16", "--f16-overflow", "fallback"])
    eng = made["m"].transformer.engine
    assert eng.f16 and made["m"].model_config["operands"] == "fp16" and made["m"].model_config["f16_overflow"] == "fallback"
    assert eng.f16_overflow_poll(sync=True) == 0
    lat16 = torch.load(os.path.join(out16, files[1]))
    d = float((lat16.double() - lat.double()).norm() / lat.double().norm())
    assert 1e-6 < d < 2e-2, d
