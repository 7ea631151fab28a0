# round 3, call a: fp8 error budget at full depth + attn_fp8-only bench line + a same-box bf16 reference line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 1500 python tools/fp8_ablation.py --out $O/fp8_ablation.json > $O/fp8_ablation.log 2>&1; echo "ablation rc=$?"
tail -40 $O/fp8_ablation.log
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc=$?"; cat $O/bench_bf16.json
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --attn-fp8 > $O/bench_attnfp8.json 2> $O/bench_attnfp8.err; echo "attnfp8 rc=$?"; cat $O/bench_attnfp8.json
