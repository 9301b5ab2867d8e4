cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rtmp; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload c4 --steps 40 --warmup 10 --cpu-steps 0 --no-extras > $O/prof.json 2> $O/prof.err
python - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/rtmp/prof/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('embed_fwd_kernel')]
s, e = idx[len(idx)//2], idx[len(idx)//2 + 1]
t0 = int(rows[s]['Start_Timestamp']); prev = t0
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f gap %6.1f dur %7.1f  grid=%-8s %s" % ((st-t0)/1e3, (st-prev)/1e3, (en-st)/1e3, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Kernel_Name'][:60]))
    prev = en
print("span %.1f us" % ((prev - t0)/1e3))
PY
find $O/prof -name '*kernel_trace.csv' -delete
tail -c 400 $O/prof.json
