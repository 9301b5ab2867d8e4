set -e
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -1
for e in "PS_WG3_LAST=0" "PS_WG3_LAST=1" "PS_WG3_LAST=0" "PS_WG3_LAST=1"; do
  env $e timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
bash tools/dbg/c2_timeline.sh
cut -c1-130 gpurun_out/c2_tl/step_timeline.txt | tail -12
