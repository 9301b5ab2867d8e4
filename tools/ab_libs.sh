# gpurun -- bash tools/ab_libs.sh [n]  : the C2 bench alternating between the diagnostic library (variant A) and the shipped one (B)
n=${1:-3}
B="python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-also"
for i in $(seq $n); do
  PS_DIAG_LIB=1 timeout -k 10 200 $B > gpurun_out/ab_a.txt 2>&1 || exit 1
  timeout -k 10 200 $B > gpurun_out/ab_b.txt 2>&1 || exit 1
  python - <<'PY'
import json
r=[]
for f in ('gpurun_out/ab_a.txt','gpurun_out/ab_b.txt'):
    d=json.loads([x for x in open(f) if x.startswith('{')][-1]); r.append((d['ms_per_step'], d['median_ms_per_step']))
print('A(diag lib) ms %.4f median %.4f | B(shipped lib) ms %.4f median %.4f' % (r[0]+r[1]), flush=True)
PY
done
