"""Drop-in alias for the reference's src/flux/generate.py -> loongx_amd.flux.generate (MI355X)."""
from loongx_amd.flux.generate import *  # noqa: F401,F403
from loongx_amd.flux import generate as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
