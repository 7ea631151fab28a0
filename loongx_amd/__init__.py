"""loongx_amd -- MI355X-native (gfx950) implementation of the LoongX denoise hot path.

Flux DiT forward (three-stream double/single blocks, QK-RMSNorm, RoPE, joint attention, AdaLN MLP) +
CS3 neural-signal encoders + DGF fusion, as hand-written HIP kernels behind a C ABI (include/lx.h),
with a Python host side that mirrors the reference's `src.flux` / `src.train.model` surface.
"""
__version__ = "0.1.0"
