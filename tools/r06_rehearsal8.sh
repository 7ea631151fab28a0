#!/usr/bin/env bash
# round 6: the 8-rank control flow at FULL size on one MI355X (eight ranks on device 0 over gloo), and the single-process replay of the
# eight ranks' seeds it has to match bit for bit
mkdir -p gpurun_out/r06b
cd "$GRAFT_REPO_ROOT"
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 1 --warmup 0 --emulate-ranks 8 --hash-latents --no-secondary --no-parity --no-cpu-baseline --no-roofline-events \
  > gpurun_out/r06b/single.out 2> gpurun_out/r06b/single.err; echo "single rc=$? wall=$(( $(date +%s) - S ))" >> gpurun_out/r06b/rc.txt
S=$(date +%s)
LX_DIST_ONE_DEVICE=1 LX_DIST_BACKEND=gloo timeout 1800 python bench.py --gpus 8 --steps 1 --warmup 0 --no-secondary --no-parity --hash-latents --no-roofline-events \
  > gpurun_out/r06b/world8.out 2> gpurun_out/r06b/world8.err; echo "world8 rc=$? wall=$(( $(date +%s) - S ))" >> gpurun_out/r06b/rc.txt
cat gpurun_out/r06b/rc.txt
tail -n 1 gpurun_out/r06b/single.out | head -c 1500; echo
tail -n 1 gpurun_out/r06b/world8.out | head -c 1500; echo
tail -n 5 gpurun_out/r06b/world8.err
