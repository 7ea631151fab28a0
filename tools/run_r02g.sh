set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --fp8 > $O/bench_fp8_512.json 2> $O/err1.txt; cat $O/bench_fp8_512.json; tail -3 $O/err1.txt
timeout 600 python bench.py --no-cpu-baseline --precise > $O/bench_precise_512.json 2> $O/err2.txt; cat $O/bench_precise_512.json; tail -3 $O/err2.txt
timeout 900 python bench.py --no-cpu-baseline --hw 64 --batch 4 --steps 1 --warmup 1 > $O/bench_bf16_1024.json 2> $O/err3.txt; cat $O/bench_bf16_1024.json; tail -3 $O/err3.txt
timeout 900 python bench.py --no-cpu-baseline --hw 64 --batch 4 --steps 1 --warmup 1 --fp8 > $O/bench_fp8_1024.json 2> $O/err4.txt; cat $O/bench_fp8_1024.json; tail -3 $O/err4.txt
timeout 900 python bench.py --no-cpu-baseline --batch 16 --modalities all --steps 1 --warmup 1 > $O/bench_bf16_b16_all.json 2> $O/err5.txt; cat $O/bench_bf16_b16_all.json; tail -3 $O/err5.txt
