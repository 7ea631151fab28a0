"""Mirror of the reference's src/flux/lora_controller.py (enable_lora :5-42, set_lora_scale :45-75) on the MI355X engine.

The reference scales PEFT layers to 0 (or by a factor) for the duration of a `with` block. Here the adapter is never a
module scale: the DiT engine evaluates it as a rank-r epilogue term of the GEMM on the rows whose stream has it on
(condition stream always, image stream with model_config["latent_lora"]). The handles the mirrored API hands out
(`LxBlock`, `LxAttention`, `LxFluxTransformer`) all index into one `DiTEngine`, and these context managers drive that
engine's switch: inside `with enable_lora(handles, False)` every block-level call made through those handles runs the
base weights on all streams (adapter term and LoRA modulation skipped), and `set_lora_scale(handles, s)` multiplies the
adapter term by s. Objects that are not engine handles are ignored, as the reference ignores non-PEFT modules.
"""
from __future__ import annotations

from typing import Any, List


def _engines(lora_modules: List[Any]):
    seen, out = set(), []
    for m in lora_modules:
        eng = getattr(m, "engine", None)
        if eng is not None and hasattr(eng, "lora_scale") and id(eng) not in seen:
            seen.add(id(eng))
            out.append(eng)
    return out


class enable_lora:
    def __init__(self, lora_modules: List[Any], activated: bool) -> None:
        self.activated = bool(activated)
        self.engines = [] if self.activated else _engines(lora_modules)
        self._saved: List[float] = []

    def __enter__(self) -> None:
        if self.activated:
            return
        self._saved = [e.lora_scale for e in self.engines]
        for e in self.engines:
            e.set_lora_scale(0.0)

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        if self.activated:
            return
        for e, s in zip(self.engines, self._saved):
            e.set_lora_scale(s)


class set_lora_scale:
    def __init__(self, lora_modules: List[Any], scale: float) -> None:
        self.engines = _engines(lora_modules)
        self.scale = float(scale)
        self._saved: List[float] = []

    def __enter__(self) -> None:
        self._saved = [e.lora_scale for e in self.engines]
        for e in self.engines:
            e.set_lora_scale(e.lora_scale * self.scale)

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        for e, s in zip(self.engines, self._saved):
            e.set_lora_scale(s)
