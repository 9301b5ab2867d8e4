for e in "PS_WGRAD_BLOCKS=512" "PS_WGRAD_BLOCKS=384" "PS_WGRAD_BLOCKS=640" "PS_WGRAD_BLOCKS=768 PS_WGRAD_ROWS=384" "PS_WGRAD_BLOCKS=512" "PS_WGRAD_BLOCKS=384"; do
  env $e timeout -k 10 120 python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
