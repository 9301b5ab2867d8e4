for e in "PS_SIDE_LIGHT=0" "PS_SIDE_LIGHT=1" "PS_X=1" "PS_SIDE_LIGHT=0" "PS_X=1"; do
  env $e timeout -k 10 200 python bench.py --workload c5 --items 8000000 --steps 50 --warmup 5 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c5 $e', d['ms_per_step'])"
done
