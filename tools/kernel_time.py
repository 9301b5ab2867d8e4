"""Average duration of kernels matching a substring, from a rocprofv3 kernel-trace directory."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
pat = sys.argv[2]
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
d = d[len(d) // 4:]
print("%s: n=%d avg=%.1f us min=%.1f us" % (pat, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3))
