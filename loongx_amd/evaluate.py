"""Evaluator of the reference's test.py (:17-214): L1 / L2 pixel distance, CLIP-I, DINO and CLIP-T over (generated, ground-truth)
image pairs -- SURVEY 8f.4, the step after the pipeline. Same function names, arguments, aggregation (mean over pairs) and
per-image result dictionaries; host-side code (PIL / numpy / torch), no kernels. The feature extractors are the caller's objects:
a `transformers` CLIPModel + CLIPProcessor from a LOCAL directory (no hub access here; the reference also loads from a local
snapshot, test.py:273-277) and, for DINO, any callable model + preprocessing function (the reference pulls `dino_vits16` through
torch.hub, test.py:286-294, which needs the network). torchvision is not in the image, so its transforms are restated below.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch
from PIL import Image


def to_tensor(img: Image.Image) -> torch.Tensor:
    """torchvision.transforms.ToTensor: HWC uint8 -> CHW float in [0, 1]."""
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()


def dino_preprocess(img: Image.Image) -> torch.Tensor:
    """Resize(256, bicubic) -> CenterCrop(224) -> ToTensor -> Normalize(ImageNet) (test.py:293-298), with torchvision's rounding:
    the longer edge becomes int(256 * long / short), the crop starts at int(round((edge - 224) / 2.0))."""
    w, h = img.size
    nw, nh = (256, int(256 * h / w)) if w <= h else (int(256 * w / h), 256)
    img = img.resize((nw, nh), resample=Image.BICUBIC)
    w, h = img.size
    l, t = int(round((w - 224) / 2.0)), int(round((h - 224) / 2.0))
    x = to_tensor(img.crop((l, t, l + 224, t + 224)))
    mean, std = torch.tensor([0.485, 0.456, 0.406])[:, None, None], torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    return (x - mean) / std


def _feat(x) -> torch.Tensor:
    """get_image_features / get_text_features return a tensor (transformers 4) or a model output with .pooler_output (5.x)."""
    return x if isinstance(x, torch.Tensor) else x.pooler_output


def _cos(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    s = float((a @ b) / (a.norm() * b.norm()))
    if s > 1 + 1e-6 or s < -1 - 1e-6:
        raise ValueError("strange similarity value")
    return s


def eval_distance(image_pairs: Sequence[Tuple[str, str]], metric: str = "l1"):
    """Mean absolute ('l1') or mean squared ('l2') pixel distance, generated image resized to the ground truth's size."""
    if metric not in ("l1", "l2"):
        raise ValueError(f"metric must be 'l1' or 'l2', got {metric}")
    total, results = 0.0, {}
    for gen_p, gt_p in image_pairs:
        gt = Image.open(gt_p).convert("RGB")
        gen = Image.open(gen_p).convert("RGB").resize(gt.size)
        d = to_tensor(gen) - to_tensor(gt)
        score = float(d.abs().mean() if metric == "l1" else (d * d).mean())
        total += score
        results.setdefault(os.path.basename(gen_p), {})[metric] = score
    return total / len(image_pairs), results


def eval_clip_i(args, image_pairs, model, processor, metric: str = "clip_i"):
    """Cosine similarity of the CLIP image features (or, metric='dino', of model(pixel_values)) of each pair."""
    def encode(image):
        inp = processor(images=image, return_tensors="pt").to(args.device)
        with torch.no_grad():
            f = _feat(model.get_image_features(inp.pixel_values)) if metric == "clip_i" else model(inp.pixel_values)
        return f.detach().cpu().float()
    total, results = 0.0, {}
    for gen_p, gt_p in image_pairs:
        s = _cos(encode(Image.open(gen_p).convert("RGB")), encode(Image.open(gt_p).convert("RGB")))
        total += s
        results.setdefault(os.path.basename(gen_p), {})[metric] = s
    return total / len(image_pairs), results


def eval_dino_i(args, image_pairs, model, processor: Callable[[Image.Image], torch.Tensor] = dino_preprocess, metric: str = "dino"):
    def encode(image):
        with torch.no_grad():
            return model(processor(image).unsqueeze(0).to(args.device)).detach().cpu().float()
    total, results = 0.0, {}
    for gen_p, gt_p in image_pairs:
        s = _cos(encode(Image.open(gen_p).convert("RGB")), encode(Image.open(gt_p).convert("RGB")))
        total += s
        results.setdefault(os.path.basename(gen_p), {})[metric] = s
    return total / len(image_pairs), results


def eval_clip_t(args, image_pairs, model, processor, caption_dict: List[Dict]):
    """Cosine similarity between each image's CLIP embedding and the CLIP text embedding of its edit instruction
    (caption_dict: the JSONL records, matched on `target_image` ending with the ground-truth file's stem)."""
    def enc_img(image):
        inp = processor(images=image, return_tensors="pt").to(args.device)
        with torch.no_grad():
            return _feat(model.get_image_features(inp.pixel_values)).detach().cpu().float()

    def enc_txt(text):
        inp = processor(text=text, return_tensors="pt", padding=True, truncation=True, max_length=77).to(args.device)
        with torch.no_grad():
            return _feat(model.get_text_features(inp.input_ids)).detach().cpu().float()
    gen_t = gt_t = 0.0
    results = {}
    for gen_p, gt_p in image_pairs:
        stem = os.path.basename(gt_p).split(".")[0]
        cap = next((it["instruction"] for it in caption_dict
                    if it["target_image"].endswith(f"{stem}.jpg") or it["target_image"].endswith(f"{stem}.png")), None)
        if cap is None:
            raise KeyError(f"no caption found for {os.path.basename(gt_p)}")
        t = enc_txt(cap)
        g = _cos(enc_img(Image.open(gen_p).convert("RGB")), t)
        gen_t += g
        gt_t += _cos(enc_img(Image.open(gt_p).convert("RGB")), t)
        results.setdefault(os.path.basename(gen_p), {})["clip-t"] = g
    return gen_t / len(image_pairs), gt_t / len(image_pairs), results


def collect_pairs(generated_path: str, gt_path: str) -> List[Tuple[str, str]]:
    """(generated, ground truth) by file name, `_0` -> `_1` (test.py:240-247)."""
    pairs = []
    for name in sorted(os.listdir(generated_path)):
        if name.endswith((".png", ".jpg")):
            gt = os.path.join(gt_path, name.replace("_0", "_1"))
            if os.path.exists(gt):
                pairs.append((os.path.join(generated_path, name), gt))
    return pairs
