"""Drop-in alias for the reference's src/flux/condition.py -> loongx_amd.flux.condition (MI355X)."""
from loongx_amd.flux.condition import *  # noqa: F401,F403
from loongx_amd.flux import condition as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
