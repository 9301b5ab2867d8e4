set -e
python -m pytest tests/test_gpu_parity_rtm.py tests/test_gpu_fullsize_rtm.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'])"
bash tools/dbg/rtm_timeline.sh > /dev/null
bash tools/dbg/rtm_alone.sh > /dev/null
