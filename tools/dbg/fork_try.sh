for e in "PS_RTM_SBWD_SIG=0" "PS_RTM_SBWD_SIG=1" "PS_RTM_SBWD_SIG=0" "PS_RTM_SBWD_SIG=1"; do
  env $e timeout -k 10 120 python bench.py --workload c4 --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c4 $e', d['ms_per_step'])"
done
