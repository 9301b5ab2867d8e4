set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_rtm.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -2
PS_GRAPHS=1 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -1
for i in 1 2 3; do
python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['ms_per_step'])"
done
python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['ms_per_step'])"
bash tools/dbg/c2_alone.sh
