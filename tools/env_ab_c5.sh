# gpurun -- bash tools/env_ab_c5.sh "ENV_A" "ENV_B" ... : same-box round-robin of environments on the C5 shard step (8 M-row table)
O=gpurun_out/env_ab; mkdir -p $O; : > $O/ab_c5.txt
one() { env $1 python bench.py --workload c5 --items 8000000 --steps 100 --warmup 10 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('median_ms_per_step'))" >> $O/ab_c5.txt; }
for i in 1 2; do for e in "$@"; do one "$e"; done; done
cat $O/ab_c5.txt
