# gpurun -- bash tools/env_ab.sh "ENV_A" "ENV_B" [bench args] : same-box alternating A/B of two environments (3 pairs of bench.py --steps 300)
A="$1"; B="$2"; shift 2
O=gpurun_out/env_ab; mkdir -p $O; : > $O/ab.txt
one() { env $1 python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-also "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; l=d.get('roofline_longest_kernel') or {}; print('$1', d['ms_per_step'], d.get('median_ms_per_step'), 'roofline-kernel us', r.get('us_per_launch'), 'longest us', l.get('us_per_launch'))" >> $O/ab.txt; }
for i in 1 2 3; do one "$A" "$@"; one "$B" "$@"; done
cat $O/ab.txt
