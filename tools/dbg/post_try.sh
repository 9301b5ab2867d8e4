set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_c5_shard.py tests/test_gpu_rowsparse.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_rtm.py -q -x 2>&1 | tail -2
for e in "PS_WGRAD_ATOMIC=1" "X=1" "PS_WGRAD_ATOMIC=1" "X=1"; do
env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 $e', d['ms_per_step'])"
done
for e in "PS_WGRAD_ATOMIC=1" "X=1"; do
env $e python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $e', d['ms_per_step'])"
env $e python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $e', d['ms_per_step'])"
done
