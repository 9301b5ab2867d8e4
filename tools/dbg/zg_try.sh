set -e
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for k in 1 0; do
PS_KEEP_GRADS=$k python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 keep $k', d['ms_per_step'])"
PS_KEEP_GRADS=$k python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 keep $k', d['ms_per_step'])"
done
