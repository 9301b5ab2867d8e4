for i in 1 2; do
for e in "PS_ATTN_WF=0" "X=1"; do
env $e python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $e', d['ms_per_step'])"
done
done
