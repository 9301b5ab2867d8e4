set -e
O=gpurun_out/x3; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --workload c5 --items 8000000 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/c5_$n.json 2>$O/c5_$n.err
  python -c "import json;d=json.loads(open('$O/c5_$n.json').read().strip().splitlines()[-1]);print('c5 $n', d['ms_per_step'])"
}
run x0 PS_GEMM_X3=0
run x1 PS_GEMM_X3=1
run x1_b1024 PS_GEMM_X3=1 PS_WGRAD_BLOCKS=1024
run x1_b2048 PS_GEMM_X3=1 PS_WGRAD_BLOCKS=2048
run x1_b2048_t22 PS_GEMM_X3=1 PS_WGRAD_BLOCKS=2048 PS_GEMM_X3_T22=384
run x1_t11 PS_GEMM_X3=1 PS_GEMM_X3_T21=100000
PS_GEMM_X3=1 python bench.py --workload c4 --steps 50 --warmup 10 --cpu-steps 0 --no-extras > $O/c4_x1.json 2>$O/c4_x1.err
python -c "import json;d=json.loads(open('$O/c4_x1.json').read().strip().splitlines()[-1]);print('c4 x1', d['ms_per_step'])"
