"""CS3 encoders + DGF fusion at BASELINE configs[2]/[3] shape (batch 16, all four modalities): the HBM-bound half of the north
star. Prints one JSON line: GPU ms per batch (HIP events), per-stage ms, algorithmic bytes and the CPU oracle's time
for the same batch on this box's host cores. Run it under rocprofv3 for per-kernel durations and
FETCH/WRITE counters:   rocprofv3 --kernel-trace --stats -d <dir> -- python tools/cs3_dgf_bench.py --iters 5 --no-cpu
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from loongx_amd.train.model import CS3DGF, synthetic_cs3_state_dict
    dev = "cuda"
    B = a.batch
    sd = synthetic_cs3_state_dict(0)
    m = CS3DGF(sd, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    eeg, fnirs, ppg, motion = r(B, 4, 4096), r(B, 6, 512), r(B, 4, 256), r(B, 6, 128)
    pe, pooled = r(B, 512, 4096) * 0.1, r(B, 768)

    stages = {}

    def timed(name, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); out = fn(); e.record()
        stages.setdefault(name, []).append((s, e))
        return out

    def run():
        ef = timed("eeg_encoder", lambda: m.eeg_projection(eeg))
        pf = timed("ppg_encoder", lambda: m.ppg_projection(ppg))
        ff = timed("fnirs_encoder", lambda: m.fnirs_projection(fnirs))
        mf = timed("motion_encoder", lambda: m.motion_projection(motion))
        pb = timed("fuse_eeg (DUAN C=512 L=4096 + Linear 1024->512)", lambda: m.fuse_eeg(ef, pf))
        qb = timed("fuse_fnirs (DUAN C=1 + Linear 1536->768)", lambda: m.fuse_fnirs(ff, mf))
        o1 = timed("duan_norm_prompt (C=512 L=4096)", lambda: m.duan_norm_prompt(pe, pb))
        o2 = timed("duan_norm_pooled (C=1 L=768)", lambda: m.duan_norm_pooled(pooled.unsqueeze(1), qb.unsqueeze(1)))
        return o1, o2

    run(); torch.cuda.synchronize()
    stages.clear()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters * 1e3
    per = {k: round(sum(s.elapsed_time(e) for s, e in v) / len(v), 4) for k, v in stages.items()}
    # algorithmic bytes: DUAN min traffic = read x, c + write y = 3 * B*C*L*4; EEG head Linear(16384->2048) streams 134 MB of fp32 weights
    duan_big = 3 * B * 512 * 4096 * 4
    rec = {"workload": f"CS3 (EEG+PPG+fNIRS+Motion encoders) + DGF fusion, batch {B}, fp32", "gpu_ms_per_batch_wall": round(wall, 3),
           "gpu_ms_per_stage": per, "gpu_ms_sum_of_stages": round(sum(per.values()), 3),
           "algorithmic_MB": {"duan C=512 L=4096 (x3: fuse_eeg, prompt)": round(duan_big / 1e6, 1),
                              "eeg head weights (16384x2048 + 2048x4096 fp32)": round((16384 * 2048 + 2048 * 4096) * 4 / 1e6, 1),
                              "eeg head output [B,512,4096] fp32": round(B * 512 * 4096 * 4 / 1e6, 1)}}
    for name in ("fuse_eeg (DUAN C=512 L=4096 + Linear 1024->512)", "duan_norm_prompt (C=512 L=4096)"):
        rec.setdefault("GBps_vs_algorithmic", {})[name] = round(duan_big / (per[name] * 1e-3) / 1e9, 1)
    if not a.no_cpu:
        from oracle import cs3 as ocs3
        torch.manual_seed(0)
        ref = ocs3.CS3DGF(seed=0).eval()
        # the same rule as bench.py's cpu_baseline (round 5): the thread count is chosen by a sweep on a warmed pool, not assumed to be
        # every hardware thread (256 threads ran this oracle 9000x slower than the GPU in rounds 2-4: mostly thread-pool hand-offs)
        c = lambda t: t.cpu()
        sweep = {}
        with torch.no_grad():
            for th in [t for t in (16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)] or [os.cpu_count() or 1]:
                torch.set_num_threads(th)
                ref.brain_embeds(c(pe[:2]), c(pooled[:2]), c(eeg[:2]), c(fnirs[:2]), c(ppg[:2]), c(motion[:2]), fuse_flag=True)      # warm this pool size
                t0 = time.time()
                ref.brain_embeds(c(pe[:2]), c(pooled[:2]), c(eeg[:2]), c(fnirs[:2]), c(ppg[:2]), c(motion[:2]), fuse_flag=True)
                sweep[th] = round((time.time() - t0) * 1e3, 1)
            best = min(sweep, key=sweep.get)
            torch.set_num_threads(best)
            ref.brain_embeds(c(pe), c(pooled), c(eeg), c(fnirs), c(ppg), c(motion), fuse_flag=True)                                  # untimed pass at the full batch
            t0 = time.time()
            ref.brain_embeds(c(pe), c(pooled), c(eeg), c(fnirs), c(ppg), c(motion), fuse_flag=True)
            rec["cpu_oracle_ms_per_batch"] = round((time.time() - t0) * 1e3, 1)
        rec["cpu_threads"], rec["cpu_threads_available"], rec["cpu_thread_sweep_ms_batch2"] = best, os.cpu_count(), {str(k): v for k, v in sweep.items()}
        rec["gpu_over_cpu"] = round(rec["cpu_oracle_ms_per_batch"] / wall, 1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
