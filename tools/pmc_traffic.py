"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/pmc_traffic.json: HBM-side bytes per
GEMM launch = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of a wide
coalesced read) + WRITE_SIZE, both in KiB per dispatch, launch-weighted over every lx_gemm_* dispatch. The record carries the
hash of the kernel source it was measured on; bench.py reports `traffic` only while that hash matches.
    python tools/pmc_traffic.py <fetch p_results.db> <write p_results.db> <label> > profiles/pmc_traffic.json
"""
import hashlib, json, os, re, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(path, counter):
    c = sqlite3.connect(path)
    tot, n, by = 0.0, 0, {}
    for name, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        if "lx_gemm" not in name:              # lx_gemm_kernel<..>, lx_gemm_mixed / _pair / _split / _fp8 kernels, lx_gemm4_kernel
            continue
        tot += v; n += 1
        k = re.search(r"lx_gemm\w+(<[^>]*>)?", name).group(0)
        a = by.setdefault(k, [0.0, 0]); a[0] += v; a[1] += 1
    return tot, n, {k: round(v[0] / v[1], 1) for k, v in by.items()}


# optional 4th / 5th argument: a workload key (b1_hw32 = headline | b16_hw32 = configs[2] | b4_hw64 = configs[4]'s per-GPU shape) and
# an existing pmc_traffic.json to merge the record into (`workloads[key]`); without them the record is the headline's, as before
f_tot, f_n, f_by = per_dispatch(sys.argv[1], "FETCH_SIZE")
w_tot, w_n, w_by = per_dispatch(sys.argv[2], "WRITE_SIZE")
sys.path.insert(0, ROOT)
from bench import gemm_sources_sha16      # one hash over every GEMM source file (gemm.hip, gemm8.h, gemm4.h, ...)
sha = gemm_sources_sha16()
mb = (2.0 * f_tot / max(f_n, 1) + w_tot / max(w_n, 1)) * 1024 / 1e6
if len(sys.argv) > 5:
    rec = json.load(open(sys.argv[5]))
    if rec.get("gemm_hip_sha16") != sha:
        raise SystemExit(f"{sys.argv[5]} was measured on another gemm.hip ({rec.get('gemm_hip_sha16')} != {sha})")
    rec.setdefault("workloads", {})[sys.argv[4]] = {"gemm_traffic_MB_per_launch": round(mb, 1), "source": sys.argv[3],
                                                     "fetch_KiB_per_launch_by_kernel (uncorrected)": f_by, "write_KiB_per_launch_by_kernel": w_by,
                                                     "launches": {"fetch_pass": f_n, "write_pass": w_n}}
    print(json.dumps(rec, indent=1))
    raise SystemExit(0)
print(json.dumps({"gemm_hip_sha16": sha, "gemm_traffic_MB_per_launch": round(mb, 1), "source": sys.argv[3],
                  "fetch_KiB_per_launch_by_kernel (uncorrected)": f_by, "write_KiB_per_launch_by_kernel": w_by,
                  "launches": {"fetch_pass": f_n, "write_pass": w_n}}, indent=1))
