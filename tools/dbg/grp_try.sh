set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py -q -x 2>&1 | tail -2
for g in 0 4096 100000; do
PS_WGRAD_GROUP_ROWS=$g python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 group_rows $g', d['ms_per_step'])"
PS_WGRAD_GROUP_ROWS=$g python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 group_rows $g', d['ms_per_step'])"
done
