set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 0 4; do
O=gpurun_out/eb_$dbg; rm -rf $O; mkdir -p $O
PS_NO_SIDE=1 PS_RTM_EB_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload c4 --steps 30 --warmup 5 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err || true
echo "dbg $dbg"; python - $O <<'PY'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'embed_bwd' in r['Name']: print('  ', r['Name'][:40], r['AverageNs'])
PY
rm -rf $O
done
