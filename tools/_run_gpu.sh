set -u
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gpu_tests_5.log 2>&1; tail -4 gpurun_out/r2_gpu_tests_5.log
python bench_finetune.py --iters 40 --no-cpu-baseline 2>/dev/null > gpurun_out/r2_bench_finetune_graph.json; cut -c1-120 gpurun_out/r2_bench_finetune_graph.json
python bench_finetune.py --iters 40 --no-cpu-baseline --no-graph 2>/dev/null > gpurun_out/r2_bench_finetune_eager.json; cut -c1-120 gpurun_out/r2_bench_finetune_eager.json
