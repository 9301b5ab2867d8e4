# gpurun -- bash tools/scatter_parts.sh : rocprofv3 average of embed_scatter_kernel in the C2 step with parts of the launch removed
# (diagnostic library, PS_SCATTER_PARTS bit mask: 1 FS-backward rows, 2 f_W gradient, 4 parked column sums, 8 scatter tasks; wrong results)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PS_DIAG_LIB=1
O=gpurun_out/scatter_parts; rm -rf $O; mkdir -p $O
for m in 15 1 2 4 8 14 13 11 7; do
  PS_SCATTER_PARTS=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$m -- python3 bench.py --steps 60 --warmup 10 --cpu-steps 0 --no-extras > $O/b$m.json 2>/dev/null
  python3 -c "import csv,sys,json; d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); [print('parts %2s: embed_scatter avg %.1f us min %.1f   step %.4f ms' % (sys.argv[3], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, d['ms_per_step'])) for r in csv.DictReader(open(sys.argv[1])) if 'embed_scatter' in r['Name']]" $(ls -t $(find $O/p$m -name '*kernel_stats.csv') | head -1) $O/b$m.json $m
  rm -rf $O/p$m
done
