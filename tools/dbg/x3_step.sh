set -e
O=gpurun_out/x3; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --workload c5 --items 8000000 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/c5_$n.json 2>$O/c5_$n.err
  python -c "import json;d=json.loads(open('$O/c5_$n.json').read().strip().splitlines()[-1]);print('c5 $n', d['ms_per_step'])"
}
run pf1 PS_GEMM_X3_PF=1
run pf2 PS_GEMM_X3_PF=2
run pf1_t22 PS_GEMM_X3_PF=1 PS_GEMM_X3_T22=384
run pf1_b PS_GEMM_X3_PF=1
run pf2_b PS_GEMM_X3_PF=2
PS_GEMM_X3_PF=1 timeout -k 10 600 python tools/gemm_x3_bench.py > $O/bench_pf1.log 2>&1
head -7 $O/bench_pf1.log
