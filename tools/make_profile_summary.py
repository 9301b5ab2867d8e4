"""profiles/<round>_summary.md from a rocprofv3 --kernel-trace --stats run of bench.py.
    python tools/make_profile_summary.py gpurun_out/final_prof gpurun_out/final_prof_bench.json gpurun_out/final_bench.json r01"""
import csv, glob, json, os, sys
prof, prof_json, bench_json, rnd = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = max(glob.glob(prof + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)   # newest run
rows = list(csv.DictReader(open(stats)))
pj = json.loads(open(prof_json).read().strip().splitlines()[-1])
bj = json.loads(open(bench_json).read().strip().splitlines()[-1])
out = []
out.append("# Round %s — rocprofv3 --kernel-trace --stats of `python bench.py --steps %d --warmup %d --cpu-steps 0` (MI355X)\n"
           % (rnd[1:], pj['steps'], pj['warmup']))
out.append("Command: `cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof "
           "-- python3 bench.py --steps %d --warmup %d --cpu-steps 0`\n" % (pj['steps'], pj['warmup']))
out.append("Workload: %s -> %d encoder replicas per row.\n" % (pj['config']['workload'], pj['config']['replicas_per_row']))
out.append("| kernel | calls | avg µs | share |\n|---|---|---|---|")
for r in rows:
    out.append("| `%s` | %s | %.1f | %s%% |" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
sc = [r for r in rows if 'score_fwd' in r['Name']]
rf = bj['roofline']
out.append("")
out.append("`score_fwd_wide_kernel` is the embedding-gather+score kernel of the roofline: the trace holds the in-step launches plus the "
           "launches of bench.py's event-timed loop; its average here is %.2f µs, bench.py's HIP-event figure (back-to-back "
           "launches, includes the inter-launch gap) %.2f µs/launch -> %.0f GB/s algorithmic = %.3f of 8 TB/s.\n"
           % (float(sc[0]['AverageNs']) / 1e3 if sc else float('nan'), rf['us_per_launch'], rf['achieved'], rf['frac']))
tj = json.load(open(os.path.join(root, 'profiles', 'gather_score_traffic.json')))
out.append("HBM traffic of that launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over `tools/gather_only.py`, "
           "%s_gather_score_pmc.txt): %s\n" % (rnd, tj['note']))
out.append("bench line (profiled run): `%s`\n" % json.dumps({k: pj[k] for k in ('value', 'unit', 'ms_per_step', 'n_gpus')}))
out.append("bench line (unprofiled default run, %s_bench.json): `%s`" % (rnd, json.dumps({k: bj[k] for k in ('value', 'unit', 'ms_per_step', 'n_gpus')})))
open(os.path.join(root, 'profiles', rnd + '_summary.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:12]))
