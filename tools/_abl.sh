set -e
python -m pytest tests -m gpu -x -q -k "grad or loss or fine_tune or finetune or backward" 2>&1 | tail -2
for r in 1 2; do for v in 0 1; do echo "== US_WGRAD_F16=$v"; US_WGRAD_F16=$v python bench_finetune.py --no-cpu-baseline --iters 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('finetune ms', d['value']*1e3)"; done; done
for v in 0 1; do echo "== pretrain US_WGRAD_F16=$v"; US_WGRAD_F16=$v python bench_pretrain.py 2>/dev/null | tail -1 | cut -c1-260; done
