import json, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.parity import full_depth_parity
for brain in (None, "all"):
    for ex in (False, True):
        mc = {"union_cond_attn": True, "attn_fp8": True}
        if ex: mc["attn_fp8_exp2"] = True
        r = full_depth_parity("cuda:0", steps=28, every=3, model_config=mc, brain=brain)
        print("brain", brain, "exp2" if ex else "loglin", {k: r[k] for k in ("noise_pred_relerr_mean", "noise_pred_relerr_max", "final_latent_relerr")}, flush=True)
