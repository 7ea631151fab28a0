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


_CANNY_WARNED = False


def canny_edges(raw_img, low: float = 100.0, high: float = 200.0):
    """`cv2.Canny(np.array(raw_img), 100, 200)` as an RGB PIL image (reference condition.py:72-76). cv2 when it is installed; otherwise
    the same algorithm in numpy (OpenCV's defaults: 3x3 Sobel with replicated borders, L1 gradient magnitude, for colour input the
    channel with the largest magnitude per pixel, non-maximum suppression along the quantised gradient direction with the tan 22.5 /
    67.5 degree sector test, 8-connected hysteresis between the two thresholds) -- restated from the published algorithm and NOT
    verified against cv2 (absent here): off the LoongX hot path (`condition_type="subject"`)."""
    import numpy as np
    from PIL import Image
    img = np.array(raw_img)
    try:
        import cv2
        return Image.fromarray(cv2.Canny(img, low, high)).convert("RGB")
    except ImportError:
        pass
    global _CANNY_WARNED
    if not _CANNY_WARNED:          # once per process: the fallback is a restatement that has never been compared with cv2 here
        import warnings
        warnings.warn("loongx_amd: cv2 is not installed -- condition_type='canny' uses the numpy / scipy restatement of cv2.Canny(img, 100, 200); "
                      "edge maps may differ from the reference's in isolated pixels", RuntimeWarning, stacklevel=2)
        _CANNY_WARNED = True
    try:
        from scipy import ndimage
    except ImportError as e:
        raise ImportError("condition_type='canny' needs cv2 (as the reference does) or, failing that, scipy for the restated algorithm") from e
    a = img.astype(np.int32)
    if a.ndim == 2:
        a = a[:, :, None]
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int32)
    dx = np.stack([ndimage.correlate(a[:, :, c], kx, mode="nearest") for c in range(a.shape[2])], -1)
    dy = np.stack([ndimage.correlate(a[:, :, c], kx.T, mode="nearest") for c in range(a.shape[2])], -1)
    mag_c = np.abs(dx) + np.abs(dy)
    best = mag_c.argmax(-1)[..., None]
    dx, dy = np.take_along_axis(dx, best, -1)[..., 0], np.take_along_axis(dy, best, -1)[..., 0]
    mag = np.abs(dx) + np.abs(dy)
    H, W = mag.shape
    m = np.zeros((H + 2, W + 2), np.int64)           # zero magnitude outside the image
    m[1:-1, 1:-1] = mag
    c = m[1:-1, 1:-1]
    x, y = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22 = x * 13573                                  # tan(22.5 deg) * 2^15 (+ 0.5)
    tg67 = tg22 + (x << 16)
    left, right, up, down = m[1:-1, :-2], m[1:-1, 2:], m[:-2, 1:-1], m[2:, 1:-1]
    s_neg = (dx ^ dy) < 0                             # gradient along the anti-diagonal
    ul, ur, dl, dr = m[:-2, :-2], m[:-2, 2:], m[2:, :-2], m[2:, 2:]
    horiz = (y < tg22) & (c > left) & (c >= right)
    vert = (y > tg67) & (c > up) & (c >= down)
    diag = (y >= tg22) & (y <= tg67) & np.where(s_neg, (c > ur) & (c > dl), (c > ul) & (c > dr))
    cand = (c > low) & (horiz | vert | diag)
    strong = cand & (c > high)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), np.int32))
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return Image.fromarray((keep[lab] * 255).astype(np.uint8)).convert("RGB")


def depth_map(raw_img, model: str = None):
    """condition.py:59-71: the `depth-estimation` pipeline of transformers on `LiheYoung/depth-anything-small-hf`, output as RGB.
    There is no hub access on the box: the model comes from a LOCAL directory (`LX_DEPTH_MODEL`, or the `model` argument)."""
    import os
    path = model or os.environ.get("LX_DEPTH_MODEL")
    if not path or not os.path.isdir(path):
        raise FileNotFoundError("condition type 'depth' needs the depth-anything model in a local directory: set LX_DEPTH_MODEL=<dir> "
                                "(the reference downloads LiheYoung/depth-anything-small-hf from the hub), or pass the prepared map as `condition=`")
    from transformers import pipeline
    import torch
    pipe = pipeline(task="depth-estimation", model=path, device="cuda" if torch.cuda.is_available() else "cpu")
    return pipe(raw_img.convert("RGB"))["depth"].convert("RGB")


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
        if condition_type == "canny":              # condition.py:72-76: cv2.Canny(img, 100, 200) -> RGB
            return canny_edges(raw_img)
        if condition_type == "depth":              # condition.py:59-71: transformers depth-estimation pipeline -> RGB
            return depth_map(raw_img)
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
