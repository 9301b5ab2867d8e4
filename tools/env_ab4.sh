# gpurun -- bash tools/env_ab4.sh "ENV_A" "ENV_B" "ENV_C" ... : same-box round-robin of several environments (2 rounds of bench.py --steps 300)
O=gpurun_out/env_ab; mkdir -p $O; : > $O/ab4.txt
one() { env $1 python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; l=d.get('roofline_longest_kernel') or {}; print('$1', d['ms_per_step'], d.get('median_ms_per_step'), 'roofline-kernel us', r.get('us_per_launch'), 'longest us', l.get('us_per_launch'))" >> $O/ab4.txt; }
for i in 1 2 3 4; do for e in "$@"; do one "$e"; done; done
cat $O/ab4.txt
