set -u
for w in 1 2 3 99; do
US_WINO_MIN_LEVEL=$w python bench_finetune.py --iters 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('finetune wino_min_level=$w', round(d['value']*1e3,2), 'ms/iter', d['first_losses'][:2])"
done
for w in 1 99; do
US_WINO_MIN_LEVEL=$w python bench_pretrain.py --iters 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pretrain wino_min_level=$w', round(d['ms_per_step'],1), d['ms_breakdown'])"
done
