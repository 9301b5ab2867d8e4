# (works at commit 882ad03 only: the experimental kernels it switches between were removed again — profiles/r05_c4_wreduce_notes.md)
# gpurun -- bash tools/wr_sweep_shapes.sh : the sweeping word-gradient reduce of C4 under (words per wave, passes) shapes, diagnostic library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_DIAG_LIB=1
O=gpurun_out/wr_sweep; rm -rf $O; mkdir -p $O
for sh in 4 5 6; do
  PS_RTM_WR_SHAPE=$sh rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$sh -- python3 bench.py --workload c4 --steps 40 --warmup 10 --cpu-steps 0 --no-extras > /dev/null 2>&1
  python3 -c "import csv,sys; [print('shape $sh:', r['Name'][:60], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)) for r in csv.DictReader(open(sys.argv[1])) if 'wreduce' in r['Name']]" $(ls -t $(find $O/p$sh -name '*kernel_stats.csv') | head -1)
  rm -rf $O/p$sh
done
PS_RTM_WR_SWEEP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pold -- python3 bench.py --workload c4 --steps 40 --warmup 10 --cpu-steps 0 --no-extras > /dev/null 2>&1
python3 -c "import csv,sys; [print('old:', r['Name'][:60], r['Calls'], 'avg us %.1f min %.1f max %.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)) for r in csv.DictReader(open(sys.argv[1])) if 'wreduce' in r['Name']]" $(ls -t $(find $O/pold -name '*kernel_stats.csv') | head -1)
rm -rf $O/pold
