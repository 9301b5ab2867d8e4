set -e
timeout -k 10 500 python -m pytest tests/test_gpu_parity_rtm.py tests/test_gpu_rtm_shapes.py tests/test_gpu_fullsize_rtm.py tests/test_gpu_dp.py -q -x 2>&1 | tail -1
for e in "PS_RTM_WR_SIDE=0" "PS_RTM_WR_SIDE=1" "PS_RTM_WR_SIDE=0" "PS_RTM_WR_SIDE=1"; do
  env $e timeout -k 10 120 python bench.py --workload c4 --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c4 $e', d['ms_per_step'])"
done
