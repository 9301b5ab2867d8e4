for e in "PS_GEMM_X3_TALL=0" "PS_GEMM_X3_TALL=1" "PS_GEMM_X3_TALL=2" "PS_GEMM_X3_TALL=0" "PS_GEMM_X3_TALL=1" "PS_GEMM_X3_TALL=2"; do
  env $e python bench.py --workload c4 --steps 200 --warmup 20 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c4 $e', d['ms_per_step'])"
done
