# (works at commit 882ad03 only: the experimental kernels it switches between were removed again — profiles/r05_c4_wreduce_notes.md)
# gpurun -- bash tools/wr_sweep_pmc.sh : FETCH_SIZE per launch of the word-gradient reduce under the sweep shapes (diagnostic library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_DIAG_LIB=1
O=gpurun_out/wr_pmc; rm -rf $O; mkdir -p $O
for sh in 0 3 4; do
  PS_RTM_WR_SHAPE=$sh rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$sh -- python3 bench.py --workload c4 --steps 12 --warmup 4 --cpu-steps 0 --no-extras > /dev/null 2>&1
  echo "shape $sh"; python tools/pmc_summary.py $O/f$sh | grep -E "wreduce"
  rm -rf $O/f$sh
done
PS_RTM_WR_SWEEP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fold -- python3 bench.py --workload c4 --steps 12 --warmup 4 --cpu-steps 0 --no-extras > /dev/null 2>&1
echo "old"; python tools/pmc_summary.py $O/fold | grep -E "wreduce"
rm -rf $O/fold
