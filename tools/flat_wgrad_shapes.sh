export PS_DIAG_LIB=1
B="python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-also"
for rep in 1 2; do
for e in "PS_X3_FLAT_SHAPE=0" "PS_X3_FLAT_SHAPE=1" "PS_X3_FLAT_SHAPE=1 PS_WGRAD_ROWS=256" "PS_X3_FLAT_SHAPE=1 PS_WGRAD_ROWS=384" "PS_X3_FLAT_SHAPE=2 PS_WGRAD_ROWS=256" "PS_X3_FLAT_SHAPE=2 PS_WGRAD_ROWS=128"; do
  env $e timeout -k 10 200 $B > gpurun_out/flat_tmp.txt 2>&1 || exit 1
  python - "$e" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/flat_tmp.txt') if x.startswith('{')][-1]
d=json.loads(l); r=d.get('roofline_longest_kernel') or {}
print(sys.argv[1], 'ms %.4f' % d['ms_per_step'], 'median %.4f' % d['median_ms_per_step'], 'wgrad us %.1f' % r.get('us_per_launch', -1), flush=True)
PY
done; done
