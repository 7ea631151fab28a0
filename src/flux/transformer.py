"""Drop-in alias for the reference's src/flux/transformer.py -> loongx_amd.flux.transformer (MI355X)."""
from loongx_amd.flux.transformer import *  # noqa: F401,F403
from loongx_amd.flux import transformer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
