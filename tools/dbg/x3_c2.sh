for e in "PS_SCORE_SIDE_LATE=0" "PS_SCORE_SIDE_LATE=1" "PS_SCORE_SIDE_LATE=0" "PS_SCORE_SIDE_LATE=1"; do
  env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
