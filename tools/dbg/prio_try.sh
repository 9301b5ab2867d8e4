for e in "PS_SIDE_PRIO=0" "X=1"; do
env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 $e', d['ms_per_step'])"
env $e python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $e', d['ms_per_step'])"
done
bash tools/dbg/c2_timeline.sh
