"""lx_attn4_kernel (loongx_amd/csrc/attn4.hip): the one-wave-per-SIMD, persistent form of the bounded-score attention that replaces
F.scaled_dot_product_attention + the mask / c_factor bias of attn_forward (src/flux/block.py:101-135).

The planner (lx_attn_fwd) hands it launches of at least two rounds of workgroups with short query-tile items; LX_ATTN4=1 / 0 in the
environment forces it on / off. Library switches are read once per process, so the forced arms run in subprocesses."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from loongx_amd import ops
from loongx_amd._lib import lib
B, H, lens, mode, out = %(B)d, %(H)d, %(lens)r, %(mode)r, %(out)r
NEG = float("-inf")
BIAS = {"none": None, "cfactor": [[0, 0, -0.6931], [0, 0, -0.6931], [-0.6931, -0.6931, 0]], "independent": [[0, 0, 0], [0, 0, 0], [NEG, NEG, 0]]}[mode]
dev = "cuda"; D = H * 128; M = B * sum(lens)
g = torch.Generator(device=dev).manual_seed(7)
buf = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
row0 = [B * sum(lens[:i]) for i in range(len(lens))]
vt0 = []          # V^T columns of a segment start 64-aligned
p = 0
for L in lens:
    vt0.append(p); p += ((L + 63) // 64) * 64
one = torch.ones(128, device=dev)
segs = [(row0[i], lens[i], vt0[i], one * ops.Q_LOG2_FACTOR, one, None, None) for i in range(len(lens))]
VT = torch.zeros(B, H, 128, p, dtype=torch.bfloat16, device=dev)
ops.qkv_prep_segs(buf, 2 * D, 0, D, segs, B, H, VT)
O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
kw = dict(q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED)
if BIAS is not None: kw["bias"] = BIAS
for _ in range(3):
    ops.attn_fwd(buf, buf, VT, O, **kw)
torch.cuda.synchronize()
print("KERNEL", lib.lx_attn_last_kernel())
np.save(out, O.view(torch.int16).cpu().numpy())
'''


def _run(tmp_path, tag, env, **shape):
    out = str(tmp_path / f"{tag}.npy")
    e = dict(os.environ)
    e.pop("LX_ATTN4", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, out=out, **shape)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    kern = int([l for l in r.stdout.splitlines() if l.startswith("KERNEL")][0].split()[1])
    raw = np.load(out)
    return kern, raw


def _f32(raw):
    return torch.from_numpy(raw.copy()).view(torch.bfloat16).float()


def _relerr(a, b):
    return float((a - b).norm() / b.norm())


def test_forced_kernel_passes_the_attention_suite():
    """Every attention test of test_kernels_gpu.py (segments, masks, c_factor, ragged tiles, the bounded-score contract) with the
    one-wave-per-SIMD kernel forced wherever the bounded contract allows it."""
    e = dict(os.environ, LX_ATTN4="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-k", "attention", "-x", "-q"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("mode", ["none", "cfactor", "independent"])
def test_persistent_launch_matches_one_item_per_workgroup_and_the_8wave_kernel(tmp_path, mode):
    """B = 12, H = 24 with ragged segments (3 + 2 + 1 query tiles per (batch, head) = 1728 items on 256 CUs: 6.75 rounds, items of 9
    key tiles whose last tiles are ragged): the persistent launch (K / V^T stream, rings and frame pipeline running across items) is
    BIT-identical from run to run, and agrees with the 8-wave kernel to bf16 rounding of P (different summation order of l). (Until round 5
    a switch ran the same kernel with one workgroup per item as a third arm: bit-identical, 300 runs of it on the driver's boxes.)"""
    shape = dict(B=12, H=24, lens=(520, 300, 100), mode=mode)
    k_p, o_p = _run(tmp_path, "persist", {"LX_ATTN4": "1"}, **shape)
    k_1, o_1 = _run(tmp_path, "again", {"LX_ATTN4": "1"}, **shape)
    k_8, o_8 = _run(tmp_path, "w8", {"LX_ATTN4": "0"}, **shape)
    assert (k_p, k_1, k_8) == (2, 2, 1)
    assert np.array_equal(o_p, o_1)
    assert _relerr(_f32(o_p), _f32(o_8)) < 2e-3
    assert np.isfinite(_f32(o_p).numpy()).all()


def test_planner_picks_the_kernel_by_launch_shape(tmp_path):
    """Unset LX_ATTN4: two or more rounds of short items -> lx_attn4_kernel; one round (batch 1) and long items -> the 8-wave kernel."""
    k_multi, _ = _run(tmp_path, "multi", {}, B=12, H=24, lens=(520, 300, 100), mode="none")
    k_one, _ = _run(tmp_path, "one", {}, B=1, H=24, lens=(512, 1024, 1024), mode="none")
    assert k_multi == 2 and k_one == 1


def test_single_tile_items_and_item_boundaries(tmp_path):
    """Items of ONE key tile (the generator runs two items ahead of the consumer; every frame is an item's first and last) and of two,
    on a persistent launch: against the 8-wave kernel."""
    for lens in [(40,), (64, 30)]:
        shape = dict(B=40, H=16, lens=lens, mode="none")        # 640 / 1280 items
        k4, o4 = _run(tmp_path, f"a{len(lens)}", {"LX_ATTN4": "1"}, **shape)
        k8, o8 = _run(tmp_path, f"b{len(lens)}", {"LX_ATTN4": "0"}, **shape)
        assert (k4, k8) == (2, 1)
        assert _relerr(_f32(o4), _f32(o8)) < 2e-3


def test_repeated_launches_are_bit_reproducible(tmp_path):
    """The persistent kernel has no inter-workgroup communication; its only hazards are inside a workgroup (rings, barriers). Forty
    back-to-back launches of a 6.75-round shape give one output."""
    code = CHILD.replace('for _ in range(3):\n    ops.attn_fwd(buf, buf, VT, O, **kw)',
                         'hs = set()\nimport hashlib\nfor _ in range(40):\n    O.zero_(); ops.attn_fwd(buf, buf, VT, O, **kw); hs.add(hashlib.sha256(O.view(torch.int16).cpu().numpy().tobytes()).hexdigest())\nprint("NHASH", len(hs))')
    out = str(tmp_path / "rep.npy")
    e = dict(os.environ, LX_ATTN4="1")
    r = subprocess.run([sys.executable, "-c", code % dict(root=ROOT, out=out, B=12, H=24, lens=(520, 300, 100), mode="cfactor")], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "NHASH 1" in r.stdout, r.stdout


def test_in_place_output_on_a_persistent_launch(tmp_path):
    """The engine runs attention IN PLACE (O over the q columns of Y) -- from a persistent workgroup that fetches the next item's Q under
    the current item's last frame. An item's Q tile is read only by the workgroup that later writes that item's O, and before it does:
    the in-place launch must equal the launch into a separate buffer bit for bit, forty times in a row (6.75 rounds of items)."""
    code = CHILD.replace('for _ in range(3):\n    ops.attn_fwd(buf, buf, VT, O, **kw)',
                         'ops.attn_fwd(buf, buf, VT, O, **kw)\nkw2 = dict(kw, o_col=2 * D)\nhs = set()\nimport hashlib\nkeep = buf.clone()\n'
                         'for _ in range(40):\n    buf.copy_(keep); ops.attn_fwd(buf, buf, VT, buf, **kw2)\n'
                         '    hs.add(hashlib.sha256(buf[:, 2 * D:].contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest())\n'
                         'hs.add(hashlib.sha256(O.view(torch.int16).cpu().numpy().tobytes()).hexdigest())\nprint("NHASH", len(hs))\n'
                         'assert torch.equal(buf[:, :2 * D], keep[:, :2 * D])')
    out = str(tmp_path / "inplace.npy")
    e = dict(os.environ, LX_ATTN4="1")
    r = subprocess.run([sys.executable, "-c", code % dict(root=ROOT, out=out, B=12, H=24, lens=(520, 300, 100), mode="cfactor")], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "KERNEL 2" in r.stdout and "NHASH 1" in r.stdout, r.stdout


def test_invariant_flag_keeps_shards_on_the_batch_kernel():
    """LX_ATTN_INVARIANT (what the engine sets with its batch-size-invariant GEMM plans): the kernel choice must not depend on the batch
    size of the launch -- at B = 16 the planner would otherwise move to lx_attn4_kernel, whose row sums are accumulated in another order.
    The rows of batch element 0 from a B = 16 launch equal a B = 1 launch of the same rows bit for bit, and both ran the 8-wave kernel."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd import ops
    from loongx_amd._lib import lib
    dev, H, lens, B = "cuda", 24, (128, 256, 256), 16
    D = H * 128
    g = torch.Generator(device=dev).manual_seed(3)
    one = torch.ones(128, device=dev)

    def run(bufs, Bn, flags):
        M = Bn * sum(lens)
        buf = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
        row0 = [Bn * sum(lens[:i]) for i in range(len(lens))]
        for s_, L_ in enumerate(lens):                      # stream-major rows: batch b of segment s at row0[s] + b * L
            buf[row0[s_]: row0[s_] + Bn * L_] = bufs[s_][: Bn * L_]
        vt0, p = [], 0
        for L_ in lens:
            vt0.append(p); p += ((L_ + 63) // 64) * 64
        VT = torch.zeros(Bn, H, 128, p, dtype=torch.bfloat16, device=dev)
        ops.qkv_prep_segs(buf, 2 * D, 0, D, [(row0[i], lens[i], vt0[i], one * ops.Q_LOG2_FACTOR, one, None, None) for i in range(len(lens))], Bn, H, VT)
        O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
        ops.attn_fwd(buf, buf, VT, O, q_col=2 * D, k_col=0, o_col=0, B=Bn, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=flags)
        torch.cuda.synchronize()
        return [O[row0[i]: row0[i] + lens[i]].clone() for i in range(len(lens))], lib.lx_attn_last_kernel()
    bufs = [torch.randn(B * L_, 3 * D, device=dev, generator=g).to(torch.bfloat16) for L_ in lens]
    base = ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED
    o16, k16 = run(bufs, B, base | ops.ATTN_INVARIANT)
    o1, k1 = run(bufs, 1, base | ops.ATTN_INVARIANT)
    assert (k16, k1) == (1, 1)                               # LX_ATTN_KERNEL_8WAVE both times
    assert all(torch.equal(a, b) for a, b in zip(o16, o1))
    import os
    if os.environ.get("LX_ATTN4") is None:
        _, kfree = run(bufs, B, base)
        assert kfree == 2                                    # without the flag this launch shape goes to lx_attn4_kernel
