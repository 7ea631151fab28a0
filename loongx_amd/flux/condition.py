"""Mirror of the reference's src/flux/condition.py (Condition :24-138, condition_dict :10-21).

Position ids reproduce the reference exactly: (0,row,col) on the packed grid, + position_delta (default for "subject":
[0, -width/16]), then the position_scale affine (:131-136).  `latents=` is an MI355X-side extension: packed
[B, N, 64] condition tokens that bypass the VAE (outside the hot path).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .pipeline_tools import encode_images

condition_dict = {"depth": 0, "canny": 1, "subject": 4, "coloring": 6, "deblurring": 7, "depth_pred": 8, "fill": 9,
                  "sr": 10, "cartoon": 11, "eeg+fnirs": 12}

_IMAGE_TYPES = ("depth", "canny", "subject", "coloring", "deblurring", "depth_pred", "fill", "sr", "cartoon")


class Condition(object):
    def __init__(self, condition_type: str, raw_img=None, condition=None, mask=None, position_delta=None,
                 position_scale=1.0, eeg=None, fnirs=None, ppg=None, motion=None, latents: Optional[torch.Tensor] = None,
                 latent_hw: Optional[Tuple[int, int]] = None) -> None:
        self.condition_type = condition_type
        assert raw_img is not None or condition is not None or latents is not None
        if raw_img is not None:
            self.condition = self.get_condition(condition_type, raw_img)
        else:
            self.condition = condition
        self.position_delta, self.position_scale = position_delta, position_scale
        self.eeg, self.fnirs, self.ppg, self.motion = eeg, fnirs, ppg, motion
        self.latents, self.latent_hw = latents, latent_hw
        assert mask is None, "Mask not supported yet"

    def get_condition(self, condition_type: str, raw_img):
        if condition_type in ("subject",):
            return raw_img
        if condition_type == "coloring":
            return raw_img.convert("L").convert("RGB")
        if condition_type in ("fill", "cartoon"):
            return raw_img.convert("RGB")
        if condition_type == "deblurring":
            from PIL import ImageFilter
            return raw_img.convert("RGB").filter(ImageFilter.GaussianBlur(10)).convert("RGB")
        if condition_type in ("depth", "canny"):
            raise NotImplementedError(f"condition type '{condition_type}' needs an external estimator (depth model / cv2), "
                                      "which is outside the LoongX path; pass the prepared image as `condition=`")
        return getattr(self, "condition", None)

    @property
    def type_id(self) -> int:
        return condition_dict[self.condition_type]

    @classmethod
    def get_type_id(cls, condition_type: str) -> int:
        return condition_dict[condition_type]

    def encode(self, pipe) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        if self.condition_type not in _IMAGE_TYPES:
            raise NotImplementedError(f"Condition type {self.condition_type} not implemented")
        if self.latents is not None:
            tokens = self.latents.to(pipe.device)
            n = tokens.shape[1]
            h2, w2 = self.latent_hw if self.latent_hw is not None else (int(round(n ** 0.5)),) * 2
            if h2 * w2 != n:
                raise ValueError(f"latent_hw={h2}x{w2} does not match {n} condition tokens")
            ids = pipe._prepare_latent_image_ids(tokens.shape[0], 2 * h2, 2 * w2, pipe.device, torch.float32)
            width_px = w2 * 16
        else:
            tokens, ids = encode_images(pipe, self.condition)
            width_px = self.condition.size[0]
        if self.position_delta is None and self.condition_type == "subject":
            self.position_delta = [0, -width_px // 16]
        if self.position_delta is not None:
            ids[:, 1] += self.position_delta[0]
            ids[:, 2] += self.position_delta[1]
        if self.position_scale != 1.0:
            scale_bias = (self.position_scale - 1.0) / 2
            ids[:, 1] *= self.position_scale
            ids[:, 2] *= self.position_scale
            ids[:, 1] += scale_bias
            ids[:, 2] += scale_bias
        type_id = torch.ones_like(ids[:, :1]) * self.type_id
        return tokens, ids, type_id
