set -e
python -m pytest tests/test_gpu_parity_rtm.py tests/test_gpu_fullsize_rtm.py -q -x 2>&1 | tail -2
for x in 0 1; do
PS_RTM_WR_X=$x python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wr_x $x', d['ms_per_step'])"
done
for w in 1024 2048 8192; do
PS_RTM_WR_WGS=$w python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wr_x wgs $w', d['ms_per_step'])"
done
bash tools/dbg/rtm_timeline.sh > /dev/null
