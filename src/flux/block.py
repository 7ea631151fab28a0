"""Drop-in alias for the reference's src/flux/block.py -> loongx_amd.flux.block (MI355X)."""
from loongx_amd.flux.block import *  # noqa: F401,F403
from loongx_amd.flux import block as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
