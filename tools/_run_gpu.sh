set -u
for i in 1 2; do timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02_bench_B1.json 2>/dev/null; cut -c1-110 gpurun_out/r02_bench_B1.json
python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_B8.json 2>/dev/null; cut -c1-110 gpurun_out/r02_bench_B8.json
python bench.py --config 64x1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_B64.json 2>/dev/null; cut -c1-110 gpurun_out/r02_bench_B64.json
timeout -k 10 400 tools/profile_bench.sh r02_prof < /dev/null | tail -16
timeout -k 10 600 tools/pmc_collect.sh r02_pmc < /dev/null | tail -3
