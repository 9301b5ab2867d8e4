set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_rowsparse.py tests/test_gpu_c5_shard.py tests/test_gpu_dp.py -q -x 2>&1 | tail -2
for i in 1 2 3; do
python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['ms_per_step'])"
done
python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['ms_per_step'])"
