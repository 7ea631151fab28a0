"""lx_attn4_kernel (loongx_amd/csrc/attn4.hip): the one-wave-per-SIMD, persistent form of the bounded-score attention that replaces
F.scaled_dot_product_attention + the mask / c_factor bias of attn_forward (src/flux/block.py:101-135).

The planner (lx_attn_fwd) hands it launches of at least two rounds of workgroups with short query-tile items; the per-launch flags
LX_ATTN_PREFER_4WAVE / LX_ATTN_INVARIANT pin the choice (until round 5 a process-static environment switch did, and every arm here was a
subprocess). tests/test_kernels_gpu.py runs its whole attention suite a second time with LX_ATTN_PREFER_4WAVE."""
import hashlib

import pytest
import torch

pytestmark = pytest.mark.gpu
NEG = float("-inf")
BIAS = {"none": None, "cfactor": [[0, 0, -0.6931], [0, 0, -0.6931], [-0.6931, -0.6931, 0]], "independent": [[0, 0, 0], [0, 0, 0], [NEG, NEG, 0]]}
K8, K4 = 1, 2                                               # LX_ATTN_KERNEL_8WAVE / LX_ATTN_KERNEL_4WAVE


class Case:
    """q / k / v of B x H heads over ragged segments, prepared as the engine does (stream-major rows, V^T image 64-aligned per segment)."""

    def __init__(self, B, H, lens, mode):
        from loongx_amd import ops
        self.ops, self.B, self.H, self.lens = ops, B, H, lens
        dev = "cuda"
        self.D = D = H * 128
        M = B * sum(lens)
        g = torch.Generator(device=dev).manual_seed(7)
        self.buf = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
        row0 = [B * sum(lens[:i]) for i in range(len(lens))]
        vt0, p = [], 0
        for L in lens:
            vt0.append(p)
            p += ((L + 63) // 64) * 64
        one = torch.ones(128, device=dev)
        self.VT = torch.zeros(B, H, 128, p, dtype=torch.bfloat16, device=dev)
        ops.qkv_prep_segs(self.buf, 2 * D, 0, D, [(row0[i], lens[i], vt0[i], one * ops.Q_LOG2_FACTOR, one, None, None) for i in range(len(lens))], B, H, self.VT)
        self.kw = dict(q_col=2 * D, k_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0)
        if BIAS[mode] is not None:
            self.kw["bias"] = BIAS[mode]
        self.base = ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED

    def run(self, pin, O=None, o_col=0, buf=None):
        """pin: ATTN_PREFER_4WAVE | ATTN_INVARIANT | 0 -> (kernel that ran, O)"""
        buf = self.buf if buf is None else buf
        if O is None:
            O = torch.zeros(buf.shape[0], self.D, dtype=torch.bfloat16, device="cuda")
        self.ops.attn_fwd(buf, buf, self.VT, O, o_col=o_col, flags=self.base | pin, **self.kw)
        torch.cuda.synchronize()
        return self.ops.lib.lx_attn_last_kernel(), O


def _relerr(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def _sha(t):
    return hashlib.sha256(t.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("mode", ["none", "cfactor", "independent"])
def test_persistent_launch_matches_the_8wave_kernel(mode):
    """B = 12, H = 24 with ragged segments (3 + 2 + 1 query tiles per (batch, head) = 1728 items on 256 CUs: 6.75 rounds, items of 9
    key tiles whose last tiles are ragged): the persistent launch (K / V^T stream, rings and frame pipeline running across items) is
    BIT-identical from run to run, and agrees with the 8-wave kernel to bf16 rounding of P (different summation order of l). (Until round 5
    a switch ran the same kernel with one workgroup per item as a third arm: bit-identical, 300 runs of it on the driver's boxes.)"""
    c = Case(12, 24, (520, 300, 100), mode)
    ops = c.ops
    k_p, o_p = c.run(ops.ATTN_PREFER_4WAVE)
    k_1, o_1 = c.run(ops.ATTN_PREFER_4WAVE)
    k_8, o_8 = c.run(ops.ATTN_INVARIANT)
    assert (k_p, k_1, k_8) == (K4, K4, K8)
    assert torch.equal(o_p, o_1)
    assert _relerr(o_p, o_8) < 2e-3
    assert torch.isfinite(o_p.float()).all()


def test_planner_picks_the_kernel_by_launch_shape():
    """No pin: two or more rounds of short items -> lx_attn4_kernel; one round (batch 1) and long items -> the 8-wave kernel."""
    assert Case(12, 24, (520, 300, 100), "none").run(0)[0] == K4
    assert Case(1, 24, (512, 1024, 1024), "none").run(0)[0] == K8
    assert Case(1, 24, (512, 4096, 4096), "none").run(0)[0] == K8             # 1632 items of 136 key tiles: long items stay on the 8-wave kernel


def test_single_tile_items_and_item_boundaries():
    """Items of ONE key tile (the generator runs two items ahead of the consumer; every frame is an item's first and last) and of two,
    on a persistent launch: against the 8-wave kernel."""
    for lens in [(40,), (64, 30)]:
        c = Case(40, 16, lens, "none")                          # 640 / 1280 items
        k4, o4 = c.run(c.ops.ATTN_PREFER_4WAVE)
        k8, o8 = c.run(c.ops.ATTN_INVARIANT)
        assert (k4, k8) == (K4, K8)
        assert _relerr(o4, o8) < 2e-3


def test_repeated_launches_are_bit_reproducible():
    """The persistent kernel has no inter-workgroup communication; its only hazards are inside a workgroup (rings, barriers). Forty
    back-to-back launches of a 6.75-round shape give one output."""
    c = Case(12, 24, (520, 300, 100), "cfactor")
    hs = set()
    O = torch.zeros(c.buf.shape[0], c.D, dtype=torch.bfloat16, device="cuda")
    for _ in range(40):
        O.zero_()
        k, _ = c.run(c.ops.ATTN_PREFER_4WAVE, O=O)
        hs.add(_sha(O))
    assert k == K4 and len(hs) == 1


def test_in_place_output_on_a_persistent_launch():
    """The engine runs attention IN PLACE (O over the q columns of Y) -- from a persistent workgroup that fetches the next item's Q under
    the current item's last frame. An item's Q tile is read only by the workgroup that later writes that item's O, and before it does:
    the in-place launch must equal the launch into a separate buffer bit for bit, forty times in a row (6.75 rounds of items)."""
    c = Case(12, 24, (520, 300, 100), "cfactor")
    D = c.D
    k, O = c.run(c.ops.ATTN_PREFER_4WAVE)
    assert k == K4
    want = _sha(O)
    keep = c.buf.clone()
    work = c.buf.clone()
    for _ in range(40):
        work.copy_(keep)
        k, _ = c.run(c.ops.ATTN_PREFER_4WAVE, O=work, o_col=2 * D, buf=work)
        assert k == K4 and _sha(work[:, 2 * D:]) == want
        assert torch.equal(work[:, :2 * D], keep[:, :2 * D])


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
    _, kfree = run(bufs, B, base)
    assert kfree == 2                                        # without the flag this launch shape goes to lx_attn4_kernel
    with pytest.raises(Exception):                           # the two pins exclude each other
        run(bufs, 1, base | ops.ATTN_INVARIANT | ops.ATTN_PREFER_4WAVE)
