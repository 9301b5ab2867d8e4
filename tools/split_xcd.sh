# gpurun -- bash tools/split_xcd.sh : row-list weight gradients (3-D grid) with / without the XCD placement of their splits (diagnostic library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_DIAG_LIB=1
for w in "c2" "c5 --items 8000000" "c4"; do
for rep in 1 2 3; do for v in 0 1; do
  PS_SPLIT_XCD=$v timeout -k 10 300 python bench.py --workload $w --steps 200 --warmup 30 --cpu-steps 0 --no-also --reps 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$w split_xcd=$v', 'ms %.4f' % d['ms_per_step'], flush=True)"
done; done
for v in 0 1; do
  rm -rf gpurun_out/sx; PS_SPLIT_XCD=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/sx -- python3 bench.py --workload $w --steps 12 --warmup 4 --cpu-steps 0 --no-extras > /dev/null 2>&1
  echo "$w PS_SPLIT_XCD=$v FETCH_SIZE KB per launch:"; python tools/pmc_summary.py gpurun_out/sx | grep "gemm_f32_kernel<1, 1, 0, 32"
done; done
rm -rf gpurun_out/sx
