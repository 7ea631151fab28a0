"""Shared helpers for the parity tests (rebuild the seeded tiny modules the goldens were made with)."""
import os

import numpy as np
import torch

from oracle import flux_modules as fm

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = dict(num_layers=2, num_single_layers=2, heads=2, head_dim=128, in_channels=64, joint_dim=64,
            pooled_dim=32, guidance_embeds=True, lora=True)


def tiny_transformer(seed=0, **over):
    cfg = dict(TINY)
    cfg.update(over)
    tr = fm.FluxTransformer2DModel(**cfg)
    fm.init_synthetic_(tr, seed=seed, std=0.05, bias_std=0.02, norm_jitter=0.1)
    return tr.eval()


def load(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def relerr(a, b):
    a, b = a.double(), b.double()
    e = float((a - b).norm() / b.norm().clamp_min(1e-30))
    rec = os.environ.get("LX_TEST_RECORD")      # tolerance audit: every measured relative error, per test, to a JSON-lines file
    if rec:
        import inspect
        import json
        fr = inspect.stack()[1]
        with open(rec, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "line": fr.lineno, "relerr": e}) + "\n")
    return e
