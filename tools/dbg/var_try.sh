for e in "X=1" "PS_SIDE_MODE=3" "PS_SIDE_MODE=0" "PS_SIDE_MODE=1"; do
env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 $e', d['ms_per_step'])"
done
