set -e
PS_GEMM_X3_SHAPE=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "mfma_gemm or wide_products" 2>&1 | tail -1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -1
for e in "PS_GEMM_X3_FLAT_PF=1" "PS_GEMM_X3_FLAT_PF=1" "PS_GEMM_X3_FLAT_PF=1"; do
  env $e python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c2 $e', d['ms_per_step'])"
done
python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c5', d['ms_per_step'])"
PS_GEMM_KSPLIT=16 timeout -k 10 300 python tools/gemm_x3_bench.py wgrad 2>&1 | cut -c1-200
