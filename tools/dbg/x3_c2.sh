export PS_GEMM_X3_FLAT=1
for e in "PS_WGRAD_BLOCKS=512" "PS_WGRAD_BLOCKS=768 PS_WGRAD_ROWS=384" "PS_WGRAD_BLOCKS=1024 PS_WGRAD_ROWS=256" "PS_WGRAD_BLOCKS=384" "PS_WGRAD_BLOCKS=256 PS_WGRAD_ROWS=1024" "PS_WGRAD_BLOCKS=512"; do
  env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
