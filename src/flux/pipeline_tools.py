"""Drop-in alias for the reference's src/flux/pipeline_tools.py -> loongx_amd.flux.pipeline_tools (MI355X)."""
from loongx_amd.flux.pipeline_tools import *  # noqa: F401,F403
from loongx_amd.flux import pipeline_tools as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
