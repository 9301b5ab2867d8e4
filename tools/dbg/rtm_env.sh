for e in "X=1" "PS_NO_FUSE=1" "PS_NO_FUSE_BWD=1" "PS_MLP_WS=0" "PS_MLP_BWD_WS=0" "PS_MLP_X3=1"; do
env $e python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['ms_per_step'])"
done
