set -e
export PS_SIDE_SEQ0=0xffefff80
timeout -k 10 200 python bench.py --steps 600 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 wrap', d['ms_per_step'], d['final_loss'])"
timeout -k 10 200 python bench.py --workload c4 --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c4 wrap', d['ms_per_step'], d['final_loss'])"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py -q -x 2>&1 | tail -1
unset PS_SIDE_SEQ0
timeout -k 10 200 python bench.py --steps 600 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2', d['ms_per_step'], d['final_loss'])"
