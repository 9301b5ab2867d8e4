set -e
for e in "PS_WG3_SIDE=1" "PS_WG3_SIDE=0" "PS_WG3_SIDE=1" "PS_WG3_SIDE=0"; do
  env $e timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
export PS_WG3_SIDE=0
bash tools/dbg/c2_timeline.sh
cut -c1-130 gpurun_out/c2_tl/step_timeline.txt | tail -14
