set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_rowsparse.py tests/test_gpu_c5_shard.py -q -x 2>&1 | tail -3
for e in "PS_ATTN_WF=0" "X=1"; do
env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 $e', d['ms_per_step'])"
done
bash tools/dbg/c2_timeline.sh
