"""CPU-only host logic: weight packing / LoRA installation (pure torch, no kernels) and the sigma schedule helpers."""
import os

import pytest
import torch

from tests.helpers import tiny_transformer


def _cfg(tr):
    from loongx_amd.flux.weights import FluxConfig
    c = tr.config
    return FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                      attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                      pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)


def test_install_lora_equals_packing_the_wrapped_state_dict(monkeypatch):
    """base weights + install_lora(diffusers-format LoRA dict) == pack_state_dict(PEFT-wrapped dict), tensor for tensor;
    alpha keys rescale the up matrices by alpha / r; partial coverage of a fused q/k/v group and unknown modules are errors."""
    from loongx_amd.flux.weights import install_lora, lora_layout, pack_state_dict
    tr = tiny_transformer()
    sd = tr.state_dict()
    cfg = _cfg(tr)
    want = pack_state_dict(sd, cfg, "cpu")
    base = {k.replace(".base_layer.", "."): v for k, v in sd.items() if ".lora_" not in k}
    lora = {"transformer." + k.replace(".default.", "."): v for k, v in sd.items() if ".lora_" in k}
    pw = pack_state_dict(base, cfg, "cpu")
    assert not pw.lora and "mod.lora_down" not in pw.t
    assert install_lora(pw, lora) == 25
    assert set(pw.lora) == set(want.lora)
    for k in want.lora:
        assert torch.equal(pw.lora[k].down, want.lora[k].down) and torch.equal(pw.lora[k].up, want.lora[k].up), k
    for k in want.t:
        assert torch.equal(pw.t[k], want.t[k]), k
    assert set(pw.t) == set(want.t)
    fused, mods = lora_layout(cfg)
    assert len(mods) == cfg.num_layers + cfg.num_single_layers and fused[0][1][-1].endswith("attn.to_q")

    with_alpha = dict(lora)
    with_alpha["transformer.x_embedder.alpha"] = torch.tensor(2.0 * lora["transformer.x_embedder.lora_A.weight"].shape[0])
    install_lora(pw, with_alpha)
    assert torch.allclose(pw.lora["x_embedder"].up, 2.0 * want.lora["x_embedder"].up)

    partial = {k: v for k, v in lora.items() if "transformer_blocks.0.attn.to_v" not in k or "single_" in k}
    with pytest.raises(ValueError):
        install_lora(pw, partial)
    with pytest.raises(KeyError):
        install_lora(pw, {**lora, "transformer.nope.lora_A.weight": torch.zeros(4, 8), "transformer.nope.lora_B.weight": torch.zeros(8, 4)})
    with pytest.raises(ValueError):
        install_lora(pw, {"transformer.x_embedder.weight": torch.zeros(2, 2)})


def test_install_lora_keeps_the_precise_mode_residuals(monkeypatch):
    """A model packed with precise=True: install_lora must leave the same adapter residuals (Lora.down_lo, mod.lora_down_lo) that
    pack_state_dict keeps, so `load_lora(dir)` (the reference's '*lora*' checkpoint branch, inference.py:43-44, under its shipped
    dtype float32) and a packed full state dict give the same precise-mode operands."""
    from loongx_amd.flux.weights import install_lora, pack_state_dict
    tr = tiny_transformer()
    sd = tr.state_dict()
    cfg = _cfg(tr)
    want = pack_state_dict(sd, cfg, "cpu", precise=True)
    base = {k.replace(".base_layer.", "."): v for k, v in sd.items() if ".lora_" not in k}
    lora = {"transformer." + k.replace(".default.", "."): v for k, v in sd.items() if ".lora_" in k}
    pw = pack_state_dict(base, cfg, "cpu", precise=True)
    install_lora(pw, lora)
    assert pw.precise_ready and any(l.down_lo is not None for l in want.lora.values())      # the synthetic adapters are not bf16-exact
    for k, l in want.lora.items():
        assert (pw.lora[k].down_lo is None) == (l.down_lo is None), k
        assert l.down_lo is None or torch.equal(pw.lora[k].down_lo, l.down_lo), k
    assert "mod.lora_down_lo" in want.t and torch.equal(pw.t["mod.lora_down_lo"], want.t["mod.lora_down_lo"])
    assert set(pw.t) == set(want.t)
    # a model packed WITHOUT residuals gets none from install_lora either
    pw0 = pack_state_dict(base, cfg, "cpu")
    install_lora(pw0, lora)
    assert all(l.down_lo is None for l in pw0.lora.values()) and "mod.lora_down_lo" not in pw0.t


def test_sigma_schedule_matches_oracle():
    import numpy as np
    from oracle import flux_modules as fm
    from loongx_amd.flux.pipeline import FlowMatchEulerDiscreteScheduler, calculate_shift
    for n, seq in ((28, 1024), (4, 4096), (50, 256)):
        sig = np.linspace(1.0, 1 / n, n)
        mu = calculate_shift(seq, 256, 4096, 0.5, 1.15)
        assert mu == fm.calculate_shift(seq, 256, 4096, 0.5, 1.15)
        a, b = FlowMatchEulerDiscreteScheduler(), fm.FlowMatchEulerDiscreteScheduler()
        a.set_timesteps(sigmas=sig, mu=mu, device="cpu"); b.set_timesteps(sigmas=sig, mu=mu)
        assert torch.equal(a.timesteps, b.timesteps) and torch.equal(a.sigmas, b.sigmas)


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
    No product module may mention it, import it lazily, or read /root/reference."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "inference.py")]
    for pkg in ("loongx_amd", "src"):
        for d, _, fs in os.walk(os.path.join(root, pkg)):
            files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
    assert len(files) > 15
    for f in files:
        src = open(f).read()
        assert "/root/reference" not in src, f
        for node in ast.walk(ast.parse(src)):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert n.split(".")[0] not in ("oracle", "refsrc", "tests"), f"{f} imports {n}"
        assert 'import_module("oracle' not in src and '__import__("oracle' not in src, f
    # bench.py: the oracle only inside cpu_baseline() and parity_check() (the checker legs); __graft_entry__: only inside smoke()
    for fn, allowed in (("bench.py", ("cpu_baseline", "parity_check")), ("__graft_entry__.py", ("smoke",))):
        tree = ast.parse(open(os.path.join(root, fn)).read())
        for top in tree.body:
            for node in ast.walk(top):
                if isinstance(node, (ast.Import, ast.ImportFrom)):
                    mod = (node.module or "") if isinstance(node, ast.ImportFrom) else ",".join(a.name for a in node.names)
                    if mod.split(".")[0] == "oracle":
                        assert isinstance(top, ast.FunctionDef) and top.name in allowed, f"{fn}: oracle imported outside {allowed}"


def test_missing_library_fails_loudly(tmp_path):
    """No CPU / torch fallback: importing the product with the HIP library absent is an ImportError that says how to build it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LX_AMD_LIB=str(tmp_path / "nope.so"), PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", "import loongx_amd.ops"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "ImportError" in r.stderr and "liblx_amd.so not found" in r.stderr and "no CPU/torch fallback" in r.stderr


def test_bench_flop_accounting_and_power_sampler_without_a_gpu():
    """bench.py's algorithmic-flop figures and the hwmon sampler's behaviour where there is no GPU / sysfs."""
    import bench
    D, S = 3072, 2560
    full = bench.flops_per_image(1024, 1024)
    # the last single block: no MLP / output projection and (round 5, lx_attn_desc.qseg_mask) no attention queries for the 1536 text / condition rows
    per_fwd = 57 * (24.0 * S * D * D + 4.0 * S * S * D) - 2.0 * 1536 * 10 * D * D - 4.0 * 1536 * S * D
    assert abs(full - 28 * per_fwd) / full < 1e-12 and 1.04e15 < full < 1.06e15            # 1.055 PFLOP per image
    cached = bench.flops_per_image_cond_cached(1024, 1024)
    assert full / 28 < cached < full and 0.55 < cached / full < 0.65                       # 27 of 28 steps run 60 % of the rows
    ps = bench.PowerSampler(0)
    ps.start()
    assert ps.stop() is None


def test_ominimodel_constructor_forms_and_lazy_brain_modules():
    """The constructor takes the reference's arguments only (no dict-key sniffing); OminiModel.from_pipe is the explicit form for
    a pipeline object + CS3 state dict; brain modules that no state dict ever filled are built randomly initialised on first
    access with a warning (the reference's constructor state, src/train/model.py:430-462) instead of an AttributeError."""
    import types
    import warnings
    from loongx_amd.train.model import OminiModel, synthetic_cs3_state_dict
    pipe = types.SimpleNamespace(transformer=types.SimpleNamespace())
    with pytest.raises(TypeError):
        OminiModel(pipe, {}, {"union_cond_attn": True}, "cpu")
    with pytest.raises(TypeError):
        OminiModel(None, {"eeg_projection.x": torch.zeros(1)})
    m = OminiModel.from_pipe(pipe, None, {"latent_lora": True}, "cpu", lora_config={"r": 4, "lora_alpha": 8})
    assert m.flux_pipe is pipe and m.model_config == {"latent_lora": True} and m.lora_scale == 2.0 and not m._brain_ready
    with pytest.raises(AttributeError):
        m.no_such_attribute
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f1 = m.fusion1
    assert m._brain_ready and f1 is m.fusion1 and any("randomly initialised" in str(x.message) for x in w)
    assert m.eeg_projection is not None and m.duan_norm_pooled is not None
    m2 = OminiModel.from_pipe(None, synthetic_cs3_state_dict(0), {}, "cpu")
    assert m2._brain_ready and m2.flux_pipe is None


def test_runtime_switches_are_the_documented_ones():
    import os
    """Every LX_* environment variable the product reads (library: getenv / env_int; Python: os.environ) is a row of INTEGRATION.md's
    switch table, and there are at most 15 of them: one plan per launch shape is the product, A/B arms live in tools/."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for base, _, files in os.walk(os.path.join(root, "loongx_amd")):
        if "__pycache__" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                found |= set(re.findall(r'(?:getenv|env_int)\("(LX_[A-Z0-9_]+)"', src))
            elif f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                found |= set(re.findall(r'environ(?:\.get\(|\[)"(LX_[A-Z0-9_]+)"', src))
    for f in ("inference.py", "bench.py", "test.py"):
        found |= set(re.findall(r'environ(?:\.get\(|\[)"(LX_[A-Z0-9_]+)"', open(os.path.join(root, f)).read()))
    found -= {"LX_DEPTH_MODEL"}                       # a model path, not a switch (documented below the table)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    table = doc[doc.index("## Runtime switches (environment)"):]
    rows = set(re.findall(r"^\| `(LX_[A-Z0-9_]+)`", table, flags=re.M))
    assert found == rows, (sorted(found - rows), sorted(rows - found))
    assert len(found) <= 15, sorted(found)


def test_bench_line_is_bounded():
    """Round 5's single JSON line grew to 21 KB and the driver's record of the round came back `parsed: null`. The final stdout line is
    now built by bench.compact_line / bench.emit: worst case = the full round-5 records (headline + six legs with two parity records on the
    largest), explanatory strings inflated, two legs failing with long error messages -- under 6 KB, one line, JSON round trip, the contract
    fields and `roofline` / `cpu_baseline` present, every leg in `summary` with its tolerance verdict."""
    import io
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = json.load(open(os.path.join(root, "profiles", "r05fin_bench_line.json")))
    sec = r.pop("secondary")
    r.pop("summary", None)
    names = ["fp16_operands_b1", "realistic_stats_b1", "configs2_b16", "hw64_b4_bf16", "configs4_b4_attnfp8", "precise_b1"]
    legs = list(zip(names, sec))
    bench.stamp_tolerance(r["parity"], {})
    assert r["parity"]["tolerance_ok"] is True and r["parity"]["tolerance_mode"] == "bf16"
    for n, lg in legs:
        mc = {"operands": "fp16"} if n.startswith("fp16") else {"attn_fp8": True} if "fp8" in n else {}
        for v in (lg.get("parity") or {}).values():
            bench.stamp_tolerance(v, mc, n.startswith("precise"), n.startswith("realistic"))
            assert v["tolerance_ok"] is True, (n, v)
    # inflate what a future edit is most likely to inflate
    r["config"]["workload"] = r["config"]["workload"] * 4
    r["roofline"]["kernel"] = r["roofline"]["kernel"] * 5
    r["cpu_baseline"]["sample"] = r["cpu_baseline"]["sample"] * 3
    legs.append(("extra_leg_a", {"error": "RuntimeError: " + "x" * 4000, "leg": "extra_leg_a"}))
    legs.append(("extra_leg_b", {"error": "LxError: " + "y" * 4000, "leg": "extra_leg_b"}))
    buf = io.StringIO()
    sidecar = os.path.join(root, "bench_legs.json")
    had = os.path.exists(sidecar)
    bench.emit(r, legs, stream=buf)
    if not had and os.path.exists(sidecar):
        os.remove(sidecar)
    out = buf.getvalue()
    assert out.endswith("\n") and out.count("\n") == 1
    line = out.strip()
    assert len(line.encode()) <= bench.LINE_LIMIT < 8192, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "summary", "value_fp16_operands", "parity_fp16_operands"):
        assert k in d, k
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["value_fp16_operands"] == sec[0]["value"] and d["parity_fp16_operands"]["ok"] is True
    assert set(d["summary"]) == {"headline", *names, "extra_leg_a", "extra_leg_b"}
    assert all(d["summary"][n].get("tolerance_ok") is True for n in names if n != "hw64_b4_bf16")
    # a tolerance that is NOT held shows: the fp8-attention figures judged as the bf16 mode
    bad = bench.stamp_tolerance(dict(sec[4]["parity"]["512x512"]), {})
    assert bad["tolerance_ok"] is False
