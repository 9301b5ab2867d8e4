"""profiles/r04_summary.md from the installed r04 profiles (after tools/refresh_profiles_r04.sh + tools/install_profiles_r02.py r04)."""
import csv, json, os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles') + '/'


def table(f, n=18, only=None):
    rows = list(csv.DictReader(open(root + f)))
    if only:
        rows = [r for r in rows if only in r['Name']]
    out = ["| kernel | calls | avg µs | share |", "|---|---|---|---|"]
    for r in rows[:n]:
        out.append("| `%s` | %s | %.1f | %s%% |" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1000, r['Percentage']))
    return "\n".join(out)


def last(f):
    return json.loads(open(root + f).read().strip().splitlines()[-1])


def span(f):
    return [l for l in open(root + f).read().splitlines() if 'launches' in l][-1].strip()


def kavg(f, name):
    for r in csv.DictReader(open(root + f)):
        if name in r['Name']:
            return float(r['AverageNs']) / 1000
    return float('nan')


b, r, c = last('r04_bench.json'), last('r04_rtm_bench.json'), last('r04_c5_bench.json')
hb = b['roofline_hbm']['by_batch']
also = {('c5' if 'd=256' in a['config']['workload'] else 'c4'): a for a in b.get('also', [])}
g1 = kavg('r04_gather_score_kernel_stats.csv', 'score_fwd_wide_kernel')
g8 = kavg('r04_gather_score_b8192_kernel_stats.csv', 'score_fwd_wide_kernel')
mlp = kavg('r04_bench_kernel_stats.csv', 'mlp_fwd_t_kernel')
s = '''# Round 04 — summary of the committed measurements (MI355X, one GPU)

Produced by `bash tools/refresh_profiles_r04.sh` on the GPU box, then `python tools/install_profiles_r02.py r04` and
`python tools/make_profile_summary_r04.py`.  Kernel statistics are `rocprofv3 --kernel-trace --stats --output-format csv -- python3
...` summaries; counter passes (`--pmc`) ran alone, as the guide prescribes.  The GEMM work of the round: `r04_gemm_notes.md`
(+ `r04_gemm_x3d_diag.txt`, `r04_gemm_x3_bench.txt`, `r04_gemm_wgrad_ksplit.txt`); gather+score variants: `r04_gather_score_variants.jsonl`;
alternative code paths: `r04_env_matrix.txt`.

## C2 — `python bench.py` (BASELINE configs[1]: item_transformer d=128, bs 384, 20 negatives, dropout 0.1)

bench line (`r04_bench.json`): **%.0f tuples/s, %.4f ms/step**, median of 200 single steps %.4f ms (p10-p90 %.4f-%.4f), %d untimed
pre-warm steps in front of the %d warm-up steps; round 3: 31.4-33.5 M tuples/s, 0.2305-0.245 ms; round 2: 27.8 M, 0.276; round 1: 22.3 M, 0.344.
Timeline `r04_step_timeline.txt`: %s.
Roofline object: `mlp_fwd_t_kernel<2,3>`, bound `mfma`, %.1f TFLOP/s of 157.3 = **%.3f** on the in-step HIP-event duration %.1f µs
(fastest launch %.1f µs); rocprofv3's average for the kernel %.1f µs = %.3f.  Unchanged kernel: round 3's own timing-only variants
(`r03_mlp_notes.md` §4: no weight loads in the chain at all = 30.3 of 33.9 µs) rule out the weight re-stream as its bound — the chain is
VALU-issue-bound (GELU, Philox, split) — so the 64-rows-per-workgroup form the review proposed was not built; DESIGN.md 5d.
`roofline_hbm` — the stand-alone gather+score launch at the C5 shape (8 M-row table, 8 rotating index sets), 40 launches back to back
between one event pair: B = 1024: %.0f GB/s = **%.3f** of 8 TB/s (%.1f µs; %.3f with an event pair around every launch, round 3's
method), B = 8192: %.0f GB/s = **%.3f** (%.1f µs; %.3f).  rocprofv3 kernel trace of the same loop (`r04_gather_score_kernel_stats.csv`,
`r04_gather_score_b8192_kernel_stats.csv`): %.1f µs and %.1f µs per launch = %.3f / %.3f.  PMC traffic `r04_gather_score_c5_pmc.txt`.
`also`: the c5 line at the STATED 50 M-row table (%.3f ms/step, in-step gather+score %.1f µs) and the c4 line (%.4f ms/step).
Kernel statistics (`r04_bench_kernel_stats.csv`, the timed steps plus the roofline pass):

%s

## C4 — `python bench.py --workload c4` (BASELINE configs[3]: review_transformer, bs 256, K 5, R 20+30, WL 100, pvc)

bench line (`r04_rtm_bench.json`): **%.0f tuples/s, %.4f ms/step** (median %.4f); round 3: 0.388-0.398 ms; round 2: 0.502; round 1: 0.733.
Roofline object: `rtm_embed4_kernel`, bound `hbm`, %.0f GB/s of 8000 = **%.3f** on %.1f MB of algorithmic bytes (in-step %.1f µs).
Timeline `r04_rtm_step_timeline.txt` (%s); kernel statistics (`r04_rtm_kernel_stats.csv`):

%s

## C5 shard — `python bench.py --workload c5 --items 8000000` (one GPU's share of BASELINE configs[4]: d=256, bs 1024, row-sparse Adam)

bench line (`r04_c5_bench.json`): **%.0f tuples/s, %.3f ms/step** (round 3: 1.34-1.39; the 50 M-row table in the default run's `also`:
%.3f); roofline object: the gather+score launch inside the step, %.0f GB/s = %.3f of peak.  What moved it this round: the big weight
gradients on 128x128 tiles of the direct-to-LDS kernel (`gemm_x3d_kernel<1,1,0,0>` below; `r04_gemm_notes.md` §3) and the replicas'
fan-in summed by a launch of its own (`fanin_sum_kernel`) so that the K/V dX product runs over the valid-row list
(`gemm_x3_kernel<0,1,1,2,1,1,1>`: 133 -> ~50 µs).  Timeline `r04_c5_step_timeline.txt` (%s); kernel statistics
(`r04_c5_kernel_stats.csv`):

%s
''' % (b['value'], b['ms_per_step'], b['median_ms_per_step'], b['p10_p90_ms_per_step'][0], b['p10_p90_ms_per_step'][1],
       b.get('prewarm_steps', 0), b['warmup'], span('r04_step_timeline.txt'),
       b['roofline']['achieved'], b['roofline']['frac'], b['roofline']['us_per_launch'], b['roofline']['us_per_launch_min'],
       mlp, b['roofline']['flops_per_launch'] / (mlp * 1e-6) / 1e12 / 157.3,
       hb[0]['achieved'], hb[0]['frac'], hb[0]['us_per_launch'], hb[0]['frac_event_pairs'],
       hb[1]['achieved'], hb[1]['frac'], hb[1]['us_per_launch'], hb[1]['frac_event_pairs'],
       g1, g8, hb[0]['bytes_per_launch'] / (g1 * 1e-6) / 8e12, hb[1]['bytes_per_launch'] / (g8 * 1e-6) / 8e12,
       also['c5']['ms_per_step'], also['c5']['roofline']['us_per_launch'], also['c4']['ms_per_step'],
       table('r04_bench_kernel_stats.csv', 17),
       r['value'], r['ms_per_step'], r['median_ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'],
       r['roofline']['bytes_per_launch'] / 1e6, r['roofline']['us_per_launch'], span('r04_rtm_step_timeline.txt'),
       table('r04_rtm_kernel_stats.csv', 22),
       c['value'], c['ms_per_step'], also['c5']['ms_per_step'], c['roofline']['achieved'], c['roofline']['frac'],
       span('r04_c5_step_timeline.txt'), table('r04_c5_kernel_stats.csv', 24))
open(root + 'r04_summary.md', 'w').write(s)
print(s[:1200])
