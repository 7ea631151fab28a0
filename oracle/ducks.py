"""Duck-typed stand-ins for the third-party objects the reference's generate() / Condition.encode() are handed
(diffusers' FluxPipeline, AutoencoderKL, VaeImageProcessor, a PIL image) -- TEST INFRASTRUCTURE.

`oracle/make_goldens.py` drives the REAL reference functions (src/flux/generate.py:72-394, src/flux/condition.py:106-138,
src/flux/pipeline_tools.py:7-30) with these objects; the tests rebuild the same objects from the same seeds and hand them to
the product's mirrors, so the goldens pin the reference's OWN logic on both sides of the boundary. The pipeline slice below
restates diffusers==0.31.0 `FluxPipeline` (train/requirements.txt:1): check_inputs, prepare_latents, _pack_latents /
_unpack_latents / _prepare_latent_image_ids, progress_bar. The VAE and the image processor are deterministic toys: what
matters for Condition.encode is the wire format around them ((x - shift) * scale, 2x2 packing, ids), not their arithmetic.
"""
from __future__ import annotations

from contextlib import contextmanager
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import flux_modules as fm


class DuckImage:
    """What Condition.encode reads off a PIL image: `.size == (width, height)` (condition.py:127)."""

    def __init__(self, width: int, height: int, seed: int = 0):
        self.size = (width, height)
        self.seed = seed


class DuckImageProcessor:
    def preprocess(self, img: DuckImage) -> torch.Tensor:
        w, h = img.size
        g = torch.Generator().manual_seed(1000 + img.seed)
        return torch.rand(1, 3, h, w, generator=g) * 2 - 1

    def postprocess(self, image, output_type="pil"):
        return image


class DuckVAE:
    """encode: 8x8 average pool + a fixed 3->16 channel mix; decode: the adjoint-ish upsample. FLUX VAE config constants."""

    def __init__(self, seed: int = 7):
        g = torch.Generator().manual_seed(seed)
        self.mix = torch.randn(16, 3, generator=g)
        self.config = SimpleNamespace(shift_factor=0.1159, scaling_factor=0.3611)

    def encode(self, images: torch.Tensor):
        z = torch.einsum("oc,bchw->bohw", self.mix.to(images), F.avg_pool2d(images, 8))
        return SimpleNamespace(latent_dist=SimpleNamespace(sample=lambda: z))

    def decode(self, z: torch.Tensor, return_dict: bool = False):
        x = torch.einsum("oc,bohw->bchw", self.mix.to(z), F.interpolate(z, scale_factor=8, mode="nearest"))
        return (x,)


class DuckFluxPipeline:
    """The slice of diffusers 0.31.0 FluxPipeline that generate() / Condition.encode() touch."""
    vae_scale_factor = 16
    default_sample_size = 64

    def __init__(self, transformer, device="cpu", dtype=torch.float32, vae=None, image_processor=None):
        self.transformer = transformer
        self.scheduler = fm.FlowMatchEulerDiscreteScheduler()
        self.vae = vae if vae is not None else DuckVAE()
        self.image_processor = image_processor if image_processor is not None else DuckImageProcessor()
        self.device, self.dtype = torch.device(device), dtype
        self._guidance_scale, self._joint_attention_kwargs, self._interrupt, self._num_timesteps = 3.5, None, False, 0
        self.adapters_set = []

    @property
    def _execution_device(self):
        return self.device

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def joint_attention_kwargs(self):
        return self._joint_attention_kwargs

    def set_adapters(self, name):
        self.adapters_set.append(name)

    def maybe_free_model_hooks(self):
        pass

    @contextmanager
    def progress_bar(self, total=None):
        yield SimpleNamespace(update=lambda *a, **k: None)

    def check_inputs(self, prompt, prompt_2, height, width, prompt_embeds=None, pooled_prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, max_sequence_length=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    def encode_prompt(self, prompt=None, prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None, device=None,
                      num_images_per_prompt=1, max_sequence_length=512, lora_scale=None):
        assert prompt_embeds is not None, "the duck pipeline has no text encoders: pass prompt_embeds"
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device, dtype=prompt_embeds.dtype)
        return prompt_embeds, pooled_prompt_embeds, text_ids

    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        x = latents.view(batch_size, num_channels_latents, height // 2, 2, width // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        return fm.unpack_latents(latents, height, width, vae_scale_factor)

    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
        ids = torch.zeros(height // 2, width // 2, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(height // 2)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(width // 2)[None, :]
        return ids.reshape(-1, 3).to(device=device, dtype=dtype)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        height = 2 * (int(height) // self.vae_scale_factor)
        width = 2 * (int(width) // self.vae_scale_factor)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), self._prepare_latent_image_ids(batch_size, height, width, device, dtype)
        shape = (batch_size, num_channels_latents, height, width)
        noise = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        return (self._pack_latents(noise, batch_size, num_channels_latents, height, width),
                self._prepare_latent_image_ids(batch_size, height, width, device, dtype))


def generate_cases():
    """(name, fuse_flag, signals used, condition_scale) of the generate() goldens."""
    allsig = ("eeg", "fnirs", "ppg", "motion")
    return [("plain", False, (), 1.0), ("eeg_only", False, ("eeg",), 1.0), ("replace", False, allsig, 1.0),
            ("fuse", True, allsig, 1.0), ("fuse_cscale2", True, allsig, 2.0), ("fnirs_motion", True, ("fnirs", "motion"), 1.0)]


def generate_inputs(hw: int = 4, seed: int = 3):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(lat=r(1, hw * hw, 64), pe=r(1, 512, 4096) * 0.1, pooled=r(1, 768),
                eeg=r(4, 3000), ppg=r(4, 256), fnirs=r(6, 600), motion=r(6, 100))


def generate_transformer(seed: int = 4):
    tr = fm.FluxTransformer2DModel(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=4096,
                                   pooled_dim=768, guidance_embeds=True, lora=True)
    fm.init_synthetic_(tr, seed=seed, std=0.03, bias_std=0.02, norm_jitter=0.1)
    return tr.eval()


# ---- evaluator fixtures (reference test.py) ----------------------------------------------------------------------------
def tiny_clip(path: str, seed: int = 0):
    """A randomly initialised `transformers` CLIPModel + CLIPProcessor (byte-level vocabulary, 32-pixel images) saved to a local
    directory: what the evaluator's CLIP-I / CLIP-T functions consume (the reference loads a local snapshot too, test.py:276-287)."""
    from tokenizers import pre_tokenizers
    from transformers import CLIPConfig, CLIPImageProcessor, CLIPModel, CLIPProcessor, CLIPTextConfig, CLIPTokenizer, CLIPVisionConfig
    alpha = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {}
    for ch in alpha:
        vocab[ch] = len(vocab)
    for ch in alpha:
        vocab[ch + "</w>"] = len(vocab)
    for w in ("<|startoftext|>", "<|endoftext|>"):
        vocab[w] = len(vocab)
    tok = CLIPTokenizer(vocab=vocab, merges=[], model_max_length=77)
    torch.manual_seed(seed)
    eos = vocab["<|endoftext|>"]
    cfg = CLIPConfig(text_config=CLIPTextConfig(vocab_size=len(vocab), hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                max_position_embeddings=77, eos_token_id=eos, bos_token_id=vocab["<|startoftext|>"], pad_token_id=eos).to_dict(),
                     vision_config=CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=32,
                                                    patch_size=8).to_dict(), projection_dim=16)
    model = CLIPModel(cfg).eval()
    model.save_pretrained(path)
    CLIPProcessor(image_processor=CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}), tokenizer=tok).save_pretrained(path)
    return model


def tiny_dino(seed: int = 0):
    """Stand-in for the DINO ViT-S/16 backbone (torch.hub, needs the network): any module mapping [1,3,224,224] -> [1,F]."""
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 16, 16), torch.nn.Flatten(), torch.nn.Linear(4 * 14 * 14, 8)).eval()


def evaluator_images(seed: int = 0):
    """Synthetic (generated, ground-truth) pairs + captions in the L-Mind layout the reference's test.py walks: generated
    `<name>_0.png`, ground truth `<name>_1.png` (test.py:241-249), JSONL records with target_image / instruction (test.py:172-178).
    Sizes differ between the two sides (the evaluator resizes the generated image to the ground truth's size, test.py:31) and one
    ground truth is non-square with an odd margin (the DINO centre crop's rounding). The caption list holds a decoy whose
    target_image also ends with img1_1.png and comes first: the reference takes the first match."""
    import numpy as np
    rng = np.random.default_rng(seed)
    gen, gt = {}, {}
    shapes = [(40, 48), (301, 263), (64, 64)]
    for i, (h, w) in enumerate(shapes):
        a = (rng.random((h, w, 3)) * 255).astype("uint8")
        gt[f"img{i}_1.png"] = a
        b = np.clip(a.astype(int) + rng.integers(-30, 30, a.shape), 0, 255).astype("uint8")
        gen[f"img{i}_0.png"] = b[: max(32, h // 2), : max(32, w // 2)].copy()
    gen["orphan_0.png"] = gen["img0_0.png"]          # no ground truth: must be skipped
    caps = [{"source_image": "x/ximg1_0.png", "target_image": "y/ximg1_1.png", "instruction": "decoy: turn it green"}]
    caps += [{"source_image": f"x/img{i}_0.png", "target_image": f"y/img{i}_1.png", "instruction": f"make it {['red', 'blue', 'a cat'][i]}"} for i in range(3)]
    return gen, gt, caps


def write_evaluator_dirs(root: str, gen, gt, caps):
    import json
    import os
    from PIL import Image
    gdir, tdir = os.path.join(root, "gen"), os.path.join(root, "gt")
    os.makedirs(gdir, exist_ok=True); os.makedirs(tdir, exist_ok=True)
    for n, a in gen.items():
        Image.fromarray(a).save(os.path.join(gdir, n))
    for n, a in gt.items():
        Image.fromarray(a).save(os.path.join(tdir, n))
    cap = os.path.join(root, "caps.jsonl")
    with open(cap, "w") as f:
        f.write("\n".join(json.dumps(c) for c in caps))
    return gdir, tdir, cap


# ---- inference CLI fixtures (reference inference.py) -------------------------------------------------------------------
def inference_case(seed: int = 0):
    """A synthetic L-Mind-style work list for the CLI's host logic: image files, a captions JSONL mixing `speech2text`, `instruction`
    and neither (-> the default prompt), one record whose source is not an image (skipped), and a brain-data pickle with a different
    subset of EEG / FNIRS / PPG / Motion per image (reference inference.py:63-74, 124-176, 264-339)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    names = [f"s{i:02d}_0.png" for i in range(5)] + ["x_0.jpg", "y_0.jpeg"]
    images = {n: (rng.random((16, 24, 3)) * 255).astype("uint8") for n in names}
    caps = [{"source_image": "imgs/s00_0.png", "target_image": "imgs/s00_1.png", "speech2text": "spoken zero", "instruction": "typed zero"},
            {"source_image": "imgs/s01_0.png", "instruction": "typed one"},
            {"source_image": "imgs/s02_0.png"},
            {"source_image": "deep/er/s03_0.png", "speech2text": "spoken three"},
            {"source_image": "notes.txt", "instruction": "not an image"},
            {"source_image": "s04_0.png", "instruction": "typed four"},
            {"source_image": "x_0.jpg", "speech2text": "spoken x"},
            {"source_image": "y_0.jpeg", "instruction": "typed y"}]
    f32 = lambda *s: rng.standard_normal(s).astype("float32")
    brain = {"s00_0.png": {"EEG": f32(4, 300), "FNIRS": f32(6, 40), "PPG": f32(4, 30), "Motion": f32(6, 20)},
             "s01_0.png": {"EEG": f32(4, 100)},
             "s03_0.png": {"FNIRS": f32(6, 64), "Motion": f32(6, 16)},
             "x_0.jpg": {"PPG": f32(4, 30)},
             "unused.png": {"EEG": f32(4, 8)}}
    return images, caps, brain


def write_inference_case(root: str, images, caps, brain):
    import json
    import os
    import pickle
    from PIL import Image
    idir = os.path.join(root, "in")
    os.makedirs(idir, exist_ok=True)
    for n, a in images.items():
        Image.fromarray(a).save(os.path.join(idir, n))
    cap = os.path.join(root, "caps.jsonl")
    with open(cap, "w") as f:
        f.write("\n".join(json.dumps(c) for c in caps))
    pkl = os.path.join(root, "brain.pkl")
    with open(pkl, "wb") as f:
        pickle.dump(brain, f)
    return idir, cap, pkl


class GenerateRecorder:
    """Stands in for `generate` / `Condition` inside an inference module: records, per call, everything the CLI decided."""

    def __init__(self):
        self.calls = []

    def condition(self, **kw):
        from types import SimpleNamespace
        return SimpleNamespace(**kw)

    def generate(self, model, pipeline, **kw):
        from types import SimpleNamespace
        import numpy as np
        from PIL import Image
        c = kw["conditions"][0]
        sig = lambda t: None if t is None else [list(t.shape), str(t.dtype).replace("torch.", ""), round(float(t.double().sum()), 4), t.device.type]
        self.calls.append(dict(
            prompt=kw.get("prompt"), height=kw["height"], width=kw["width"], seed=int(kw["generator"].initial_seed()),
            default_lora=bool(kw["default_lora"]), fuse_flag=bool(kw["fuse_flag"]), use_brain_condition=bool(kw["use_brain_condition"]),
            model_config=dict(kw["model_config"]), pipeline_is_model_pipe=pipeline is model.flux_pipe,
            condition_type=c.condition_type, position_delta=list(c.position_delta), condition_size=list(c.condition.size),
            condition_mode=c.condition.mode, condition_sum=int(np.asarray(c.condition, dtype=np.int64).sum()),
            cond_signals=[sig(getattr(c, k)) for k in ("eeg", "fnirs", "ppg", "motion")],
            signals=[sig(kw.get(f"additional_condition{i}")) for i in (1, 2, 3, 4)]))
        n = len(self.calls)
        return SimpleNamespace(images=[Image.fromarray(np.full((8, 8, 3), n, np.uint8))])
