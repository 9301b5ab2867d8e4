set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 2 4 8 14; do
O=gpurun_out/e4_$dbg; rm -rf $O; mkdir -p $O
PS_NO_SIDE=1 PS_RTM_E4_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload c4 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err || true
echo "dbg $dbg" | tee -a gpurun_out/e4.log; python - $O <<'PY'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'embed4' in r['Name']: print('  ', r['Name'][:40], r['AverageNs'])
PY
rm -rf $O
done
