set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_c5_shard.py tests/test_gpu_rowsparse.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['ms_per_step'])"
done
for e in "PS_ATTN_WF=0" "X=1"; do
env $e python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $e', d['ms_per_step'])"
done
