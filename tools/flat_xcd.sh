# gpurun -- bash tools/flat_xcd.sh : the flat weight-gradient group with / without the XCD-aware split placement (diagnostic library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_DIAG_LIB=1
B="python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-also --reps 0"
for rep in 1 2 3; do
for e in "PS_FLAT_XCD=0" "PS_FLAT_XCD=1"; do
  env $e timeout -k 10 200 $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e', 'ms %.4f' % d['ms_per_step'], 'wgrad us %.1f' % d['roofline_longest_kernel']['us_per_launch'], flush=True)"
done; done
for v in 0 1; do
  rm -rf gpurun_out/fx; PS_FLAT_XCD=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fx -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > /dev/null 2>&1
  echo "PS_FLAT_XCD=$v FETCH_SIZE KB per launch:"; python tools/pmc_summary.py gpurun_out/fx | grep "gemm_x3_kernel<1, 1, 0, 1"
done
rm -rf gpurun_out/fx
