"""Print the kernel timeline of one profiled training step from a rocprofv3 kernel-trace CSV."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(('sample_kernel', 'tem_stage_kernel'))]
if len(idx) < 2:      # sampling folded into the first kernel of the step
    idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('embed_fwd_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
s, e = idx[k], idx[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = t0
tot = 0
for r in rows[s:e]:
    st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    print("%8.1f  gap %5.1f  dur %6.1f us  grid=%-12s wg=%-4s %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3,
          '%sx%sx%s' % (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z']), r['Workgroup_Size_X'], r['Kernel_Name'][:48]))
    prev_end = en; tot += en - st
print("step span %.1f us, kernel sum %.1f us, launches %d" % ((prev_end - t0) / 1e3, tot / 1e3, e - s))
