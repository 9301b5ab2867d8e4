"""Print the kernel timeline of one profiled training step from a rocprofv3 kernel trace (CSV output, or the rocpd database
rocprofv3 writes by default): python tools/trace_step.py <dir> [step index] [first kernel prefix]"""
import csv, glob, sqlite3, sys


def load(d):
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))
    if f:
        rows = []
        for r in csv.DictReader(open(f[0])):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                         int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Workgroup_Size_X'])))
        return sorted(rows)
    db = sqlite3.connect(sorted(glob.glob(d + '/**/*.db', recursive=True))[0])
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = ("select d.start, d.end, s.kernel_name, d.grid_size_x / max(1, d.workgroup_size_x), d.workgroup_size_x "
         "from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks))
    return [(a, b, n.replace('.kd', ''), g, w) for a, b, n, g, w in c.execute(q)]


rows = load(sys.argv[1])
first = sys.argv[3] if len(sys.argv) > 3 else None
names = (first,) if first else ('sample_kernel', 'tem_stage_kernel')
idx = [i for i, r in enumerate(rows) if any(n in r[2] for n in names)]
if len(idx) < 2:      # sampling folded into the first kernel of the step
    idx = [i for i, r in enumerate(rows) if 'embed_fwd_kernel' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
s, e = idx[k], idx[k + 1]
t0 = rows[s][0]
prev_end = t0
tot = 0
for st, en, name, grid, wg in rows[s:e]:
    print("%8.1f  gap %5.1f  dur %6.1f us  grid=%-8d wg=%-4d %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, grid, wg, name[:64]))
    prev_end = max(prev_end, en); tot += en - st
print("step span %.1f us, kernel sum %.1f us, launches %d" % ((prev_end - t0) / 1e3, tot / 1e3, e - s))
