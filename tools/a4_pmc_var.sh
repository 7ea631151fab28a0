#!/usr/bin/env bash
# Wave-cycle accounting (clock-independent: SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY, quad-cycles) of variant
# builds of lx_attn4_kernel:   tools/a4_pmc_var.sh <out name> "<attn_run.py args>" name1 name2 ...   (names of loongx_amd/lib/liblx_amd_a4<name>.so; "base")
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
ROOT=$PWD; O=$ROOT/gpurun_out/$1; ARGS=$2; shift 2; mkdir -p $(dirname $O)
cd /tmp && export TMPDIR=/tmp
: > $O
for n in "$@"; do
  lib=$ROOT/loongx_amd/lib/liblx_amd_a4$n.so; [[ "$n" == "base" ]] && lib=$ROOT/loongx_amd/lib/liblx_amd.so
  rm -rf /tmp/pm
  LX_AMD_LIB=$lib timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/pm -o p -- python $ROOT/tools/attn_run.py $ARGS > /dev/null 2>&1
  python $ROOT/tools/db_summary.py /tmp/pm/p_results.db 0.0 2>/dev/null | grep -i "attn4" | awk -v n=$n '{printf "%-12s %s\n", n, $0}' >> $O
  rm -rf /tmp/pm
  LX_AMD_LIB=$lib timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pm -o p -- python $ROOT/tools/attn_run.py $ARGS > /dev/null 2>&1
  python $ROOT/tools/db_summary.py /tmp/pm/p_results.db 0.0 2>/dev/null | grep -i "attn4" | awk -v n=$n '{printf "%-12s %s\n", n, $0}' >> $O
done
python3 - $O <<'PY'
import re, sys
rows = {}
for l in open(sys.argv[1]):
    n = l.split()[0]
    d = rows.setdefault(n, {})
    m = re.search(r"\s(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s", l)
    if m: d.setdefault("us", float(m.group(3))); d["waves"] = int(m.group(1)) * 4
    for k, v in re.findall(r"(\w+)=([\d.e+]+)", l): d[k] = float(v)
print(f"{'variant':12s} {'us':>8s} {'cyc/wave':>10s} {'active':>8s} {'inst-stall':>10s} {'parked':>8s} {'MHz':>6s} {'mfma busy':>9s}")
for n, d in rows.items():
    w = d.get("waves", 1)
    wc = 4 * d.get("SQ_WAVE_CYCLES", 0) / w
    print(f"{n:12s} {d.get('us', 0):8.1f} {wc:10.0f} {4 * d.get('SQ_ACTIVE_INST_ANY', 0) / w / max(wc, 1):8.3f} {4 * d.get('SQ_WAIT_INST_ANY', 0) / w / max(wc, 1):10.3f} "
          f"{4 * d.get('SQ_WAIT_ANY', 0) / w / max(wc, 1):8.3f} {d.get('GRBM_GUI_ACTIVE', 0) / 8 / max(d.get('us', 1), 1e-9):6.0f} {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(d.get('GRBM_GUI_ACTIVE', 1) * 128, 1):9.3f}")
PY
