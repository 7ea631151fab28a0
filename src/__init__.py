"""Alias package: the reference's import surface (`src.flux.*`, `src.train.model`) backed by loongx_amd."""
