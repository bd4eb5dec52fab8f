set -u
t0=$(date +%s)
python bench.py > gpurun_out/r02_bench_B1.json 2> gpurun_out/r02_bench_B1.err; echo "default bench: $(( $(date +%s) - t0 )) s"; cut -c1-120 gpurun_out/r02_bench_B1.json
python bench.py --config 64x1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_B64.json 2>/dev/null; cut -c1-120 gpurun_out/r02_bench_B64.json
python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_B8.json 2>/dev/null; cut -c1-120 gpurun_out/r02_bench_B8.json
python bench_finetune.py --iters 50 > gpurun_out/r02_bench_finetune.json 2>/dev/null; cut -c1-120 gpurun_out/r02_bench_finetune.json
python bench_pretrain.py --iters 5 > gpurun_out/r02_bench_pretrain.json 2>/dev/null; cut -c1-200 gpurun_out/r02_bench_pretrain.json
timeout -k 10 400 tools/profile_bench.sh r02_prof < /dev/null | tail -3
timeout -k 10 600 tools/pmc_collect.sh r02_pmc < /dev/null | tail -4
