# gpurun -- bash tools/ab_diag_slot.sh : same-box alternating A/B of two BUILDS.  The library under test is lib/libprodsearch_hip.so; the
# library to compare against is copied into the diagnostic slot (lib/libprodsearch_hip_diag.so, selected by PS_DIAG_LIB=1) before the call.
O=gpurun_out/ab_slot; mkdir -p $O; : > $O/ab.txt
one() { python bench.py $2 --cpu-steps 0 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d.get('median_ms_per_step'), (d.get('roofline') or {}).get('us_per_launch'))" >> $O/ab.txt; }
for i in 1 2 3; do
  PS_DIAG_LIB=1 one old "--steps 300 --warmup 30"
  one new "--steps 300 --warmup 30"
done
for i in 1 2; do
  PS_DIAG_LIB=1 one old "--workload c5 --items 8000000 --steps 100 --warmup 10"
  one new "--workload c5 --items 8000000 --steps 100 --warmup 10"
  PS_DIAG_LIB=1 one old "--workload c4 --steps 200 --warmup 20"
  one new "--workload c4 --steps 200 --warmup 20"
done
cat $O/ab.txt
