"""Builds a tiny LOCAL diffusers-format FLUX.1 directory (transformer/, vae/, text_encoder*/, tokenizer*/) and a LoongX-style
full state dict from seeded oracle modules -- what `OminiModel(flux_pipe_id=<dir>)` + `load_state_dict` / inference.py consume.
Test infrastructure: there are no real checkpoints or tokenizer files on the box."""
import json
import os

import torch
from safetensors.torch import save_file

from oracle import cs3 as ocs3
from oracle import flux_modules as fm
from oracle import vae as ovae

T5_DIM, CLIP_DIM = 64, 32
VAE_CFG = dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 256, 256), layers_per_block=1,
               norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159)
WORDS = "make the sky red blue a cat dog edit this image".split()


def tiny_transformer(seed=4):
    tr = fm.FluxTransformer2DModel(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=T5_DIM,
                                   pooled_dim=CLIP_DIM, guidance_embeds=True, lora=True)
    fm.init_synthetic_(tr, seed=seed, std=0.03, bias_std=0.02, norm_jitter=0.1)
    return tr.eval()


def tiny_vae(seed=5):
    return ovae.init_synthetic_(ovae.AutoencoderKL(**VAE_CFG), seed)


def build_text_parts(d):
    from tokenizers import pre_tokenizers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer, T5Config, T5EncoderModel, T5TokenizerFast
    alpha = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {}
    for ch in alpha:
        vocab[ch] = len(vocab)
    for ch in alpha:
        vocab[ch + "</w>"] = len(vocab)
    for w in ("<|startoftext|>", "<|endoftext|>"):
        vocab[w] = len(vocab)
    CLIPTokenizer(vocab=vocab, merges=[], model_max_length=77).save_pretrained(os.path.join(d, "tokenizer"))
    torch.manual_seed(11)
    eos = vocab["<|endoftext|>"]
    CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=CLIP_DIM, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                 max_position_embeddings=77, projection_dim=CLIP_DIM, eos_token_id=eos, bos_token_id=vocab["<|startoftext|>"],
                                 pad_token_id=eos)).eval().save_pretrained(os.path.join(d, "text_encoder"))
    pieces = [("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0)] + [("▁" + w, -2.0 - 0.01 * i) for i, w in enumerate(WORDS)]
    pieces += [(c, -8.0) for c in "abcdefghijklmnopqrstuvwxyz▁"]
    T5TokenizerFast(vocab=pieces, extra_ids=0).save_pretrained(os.path.join(d, "tokenizer_2"))
    T5EncoderModel(T5Config(vocab_size=len(pieces), d_model=T5_DIM, d_kv=16, d_ff=128, num_layers=2, num_heads=4,
                            feed_forward_proj="gated-gelu")).eval().save_pretrained(os.path.join(d, "text_encoder_2"))


def build_flux_dir(d, tr=None, vae=None, with_text=True):
    """-> (transformer, vae) oracle modules whose weights were written under d/."""
    tr, vae = tr or tiny_transformer(), vae or tiny_vae()
    os.makedirs(os.path.join(d, "transformer"), exist_ok=True)
    os.makedirs(os.path.join(d, "vae"), exist_ok=True)
    # the pipeline directory holds the BASE model: plain Linear names, no adapters (those come with the LoongX checkpoint)
    base = {}
    for k, v in tr.state_dict().items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        base[k.replace(".base_layer.", ".")] = v.contiguous()
    keys = sorted(base)
    half = len(keys) // 2                                    # two shards + an index, like the real checkpoint
    shards = {"diffusion_pytorch_model-00001-of-00002.safetensors": keys[:half], "diffusion_pytorch_model-00002-of-00002.safetensors": keys[half:]}
    wm = {}
    for f, ks in shards.items():
        save_file({k: base[k] for k in ks}, os.path.join(d, "transformer", f))
        wm.update({k: f for k in ks})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(d, "transformer", "diffusion_pytorch_model.safetensors.index.json"), "w"))
    c = tr.config
    json.dump(dict(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                   attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                   pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=list(c.axes_dims_rope)),
              open(os.path.join(d, "transformer", "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in vae.state_dict().items()}, os.path.join(d, "vae", "diffusion_pytorch_model.safetensors"))
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in VAE_CFG.items()}, open(os.path.join(d, "vae", "config.json"), "w"))
    if with_text:
        build_text_parts(d)
    return tr, vae


def loongx_state_dict(tr, brain=None):
    """A full LoongX checkpoint in the reference's naming: `transformer.*` (PEFT-wrapped where LoRA is attached) + the brain-side
    modules (inference.py:46-53)."""
    if brain is None:
        torch.manual_seed(0)
        brain = ocs3.CS3DGF(seed=0).eval()
    sd = {"transformer." + k: v for k, v in tr.state_dict().items()}
    sd.update(brain.state_dict())
    return sd, brain
