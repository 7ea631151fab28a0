"""Drop-in alias for the reference's src/flux/lora_controller.py -> loongx_amd.flux.lora_controller (MI355X)."""
from loongx_amd.flux.lora_controller import *  # noqa: F401,F403
from loongx_amd.flux import lora_controller as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
