# gpurun -- bash tools/c2_pmc_traffic.sh : memory-side traffic per launch of the C2 step's kernels (separate FETCH_SIZE / WRITE_SIZE passes,
# MI355X_MICROARCH.md HBM section: FETCH_SIZE x2 for 16-byte-per-lane streaming reads on gfx950; Infinity-Cache hits are counted too)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $O/w.log 2>&1 || exit 1
( echo "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras   (per-launch average, KB as reported; gfx950: x2 for 16-B/lane streaming reads)"
  python tools/pmc_summary.py $O/f
  echo; echo "rocprofv3 --pmc WRITE_SIZE -- same command   (per-launch average, KB)"
  python tools/pmc_summary.py $O/w ) > gpurun_out/r05_c2_pmc_traffic.txt
rm -rf $O/f $O/w
cat gpurun_out/r05_c2_pmc_traffic.txt | cut -c1-100
