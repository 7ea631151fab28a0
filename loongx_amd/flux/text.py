"""Prompt encoding for the FLUX pipeline: CLIP-L pooled embedding + T5-XXL sequence embedding (diffusers 0.31.0
`FluxPipeline.encode_prompt` / `_get_clip_prompt_embeds` / `_get_t5_prompt_embeds`; reference call sites
src/flux/generate.py:152-165, src/flux/pipeline_tools.py:33-52).

The two encoders are host-side `transformers` modules running on PyTorch-ROCm (they run once per image, outside the denoise
loop and outside the metric): this class owns the tokenisation conventions and the output contract
`(prompt_embeds [B, L, 4096], pooled_prompt_embeds [B, 768])` that `LxFluxPipeline(text_encoder=...)` expects. Checkpoints and
tokenizer files come from a local diffusers-format FLUX.1 directory (there is no hub access on the box).
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import torch


class FluxTextEncoders:
    def __init__(self, text_encoder, tokenizer, text_encoder_2, tokenizer_2, device="cuda", dtype=torch.bfloat16,
                 tokenizer_max_length: Optional[int] = None):
        self.text_encoder, self.tokenizer = text_encoder, tokenizer              # CLIPTextModel / CLIPTokenizer
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2      # T5EncoderModel / T5TokenizerFast
        self.device, self.dtype = torch.device(device), dtype
        self.tokenizer_max_length = tokenizer_max_length or getattr(tokenizer, "model_max_length", 77)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda", dtype=torch.bfloat16):
        from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5TokenizerFast
        need = ("text_encoder", "tokenizer", "text_encoder_2", "tokenizer_2")
        missing = [d for d in need if not os.path.isdir(os.path.join(path, d))]
        if missing:
            raise FileNotFoundError(f"{path}: missing {missing} (a diffusers-format FLUX.1 directory has all of {need})")
        te = CLIPTextModel.from_pretrained(os.path.join(path, "text_encoder"), torch_dtype=dtype).to(device).eval()
        te2 = T5EncoderModel.from_pretrained(os.path.join(path, "text_encoder_2"), torch_dtype=dtype).to(device).eval()
        return cls(te, CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer")), te2,
                   T5TokenizerFast.from_pretrained(os.path.join(path, "tokenizer_2")), device, dtype)

    @torch.no_grad()
    def clip_pooled(self, prompt: List[str]) -> torch.Tensor:
        ids = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer_max_length, truncation=True,
                             return_overflowing_tokens=False, return_length=False, return_tensors="pt").input_ids
        out = self.text_encoder(ids.to(self.device), output_hidden_states=False)
        return out.pooler_output.to(dtype=self.dtype, device=self.device)

    @torch.no_grad()
    def t5_sequence(self, prompt: List[str], max_sequence_length: int = 512) -> torch.Tensor:
        ids = self.tokenizer_2(prompt, padding="max_length", max_length=max_sequence_length, truncation=True, return_length=False,
                               return_overflowing_tokens=False, return_tensors="pt").input_ids
        return self.text_encoder_2(ids.to(self.device), output_hidden_states=False)[0].to(dtype=self.dtype, device=self.device)

    def __call__(self, prompt: Union[str, List[str]], prompt_2: Optional[Union[str, List[str]]] = None, max_sequence_length: int = 512):
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        prompt_2 = prompt if prompt_2 is None else ([prompt_2] if isinstance(prompt_2, str) else list(prompt_2))
        return self.t5_sequence(prompt_2, max_sequence_length), self.clip_pooled(prompt)
