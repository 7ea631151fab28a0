#!/usr/bin/env python3
"""A/B of attention-kernel build / environment variants in alternating subprocesses (library switches are read once per process):
    python tools/attn_ab.py [--big] [--fp8] VAR=VAL[,VAR=VAL] ...      each argument = one arm; "base" = no variables
Prints the minimum over passes of the mean launch time per arm, and a checksum of the output (arms that only change the
schedule or the store shape must agree bit for bit)."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from loongx_amd import ops
big, fp8 = %r, %r
dev = "cuda"; B, H = 1, 24; lens = (512, 4096, 4096) if big else (512, 1024, 1024)
import os
if os.environ.get("AB_SHAPE"):        # AB_SHAPE=BxHxL: one segment (e.g. 16x64x2048: the shape the programming guide quotes attention rates on)
    B, H, L0 = (int(v) for v in os.environ["AB_SHAPE"].split("x")); lens = (L0,)
D = H * 128
M = B * sum(lens)
g = torch.Generator(device=dev).manual_seed(0)
buf = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
row0 = [B * sum(lens[:i]) for i in range(len(lens))]; vt0 = [sum(lens[:i]) for i in range(len(lens))]
flags = int(os.environ.get("AB_FLAGS", "0"))          # AB_NORM=1: RMS-normalised q / k (what the engine feeds); AB_FLAGS=3: + the bounded-score kernel
one = torch.ones(128, device=dev) if (flags or os.environ.get("AB_NORM")) else None
oneq = one * ops.Q_LOG2_FACTOR if flags else one
segs = [(row0[i], lens[i], vt0[i], oneq, one, None, None) for i in range(len(lens))]
O = torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
if fp8:
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8)
    VT8 = torch.zeros(B, H, 128, sum(lens), dtype=torch.uint8, device=dev)
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
    run = lambda: ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=int(os.environ.get("AB_FLAGS8", "0")))      # (AB_FLAGS8=32: LX_ATTN_P_EXP2)
else:
    VT = torch.zeros(B, H, 128, sum(lens), dtype=torch.bfloat16, device=dev)
    q = buf.clone()
    ops.qkv_prep_segs(q, 2 * D, 0, D, segs, B, H, VT)
    run = lambda: ops.attn_fwd(q, q, VT, O, q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=flags)
for _ in range(10): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): run()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) * 1e3 / 40)
S = sum(lens)
print("RESULT", best, 4 * B * H * S * S * 128 / best / 1e6, hashlib.sha256(O.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12])
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    big, fp8 = "--big" in sys.argv, "--fp8" in sys.argv
    arms = args or ["base"]
    best = {a: (1e9, 0.0, "") for a in arms}
    for p in range(3):
        for a in arms:
            env = dict(os.environ)
            if a != "base":
                for kv in a.split(","):
                    k, v = kv.split("=")
                    env[k] = v
            r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, big, fp8)], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(a, "FAILED", r.stderr[-400:])
                continue
            _, us, tf, h = line[0].split()
            if float(us) < best[a][0]:
                best[a] = (float(us), float(tf), h)
    for a in arms:
        print(f"{a:50s} {best[a][0]:8.1f} us  {best[a][1]:7.0f} TF  out {best[a][2]}")


if __name__ == "__main__":
    main()
