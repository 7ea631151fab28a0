"""Mirror of the reference's src/flux/lora_controller.py (enable_lora :5-42, set_lora_scale :45-75).

On MI355X LoRA is never applied by mutating module scales: the DiT engine evaluates the adapter as a rank-r epilogue
term on exactly the rows (token streams) that have it enabled.  These context managers therefore only record the
decision on the handles they are given, so reference-style call sites keep working.
"""
from __future__ import annotations

from typing import Any, List


class enable_lora:
    """`with enable_lora(modules, activated)`: when not activated the adapter contributes nothing inside the block."""

    def __init__(self, lora_modules: List[Any], activated: bool) -> None:
        self.activated = bool(activated)
        self.lora_modules = [m for m in lora_modules if hasattr(m, "lora_enabled")]
        self._saved: List[bool] = []

    def __enter__(self) -> None:
        if self.activated:
            return
        self._saved = [m.lora_enabled for m in self.lora_modules]
        for m in self.lora_modules:
            m.lora_enabled = False

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        if self.activated:
            return
        for m, s in zip(self.lora_modules, self._saved):
            m.lora_enabled = s


class set_lora_scale:
    def __init__(self, lora_modules: List[Any], scale: float) -> None:
        self.lora_modules = [m for m in lora_modules if hasattr(m, "lora_scale")]
        self.scale = scale
        self._saved: List[float] = []

    def __enter__(self) -> None:
        self._saved = [m.lora_scale for m in self.lora_modules]
        for m in self.lora_modules:
            m.lora_scale = m.lora_scale * self.scale

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        for m, s in zip(self.lora_modules, self._saved):
            m.lora_scale = s
