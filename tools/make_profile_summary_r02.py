"""profiles/r02_summary.md from the installed r02 profiles (after tools/install_profiles_r02.py)."""
import csv, json, os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles') + '/'


def table(f, n=18):
    rows = list(csv.DictReader(open(root + f)))
    out = ["| kernel | calls | avg µs | share |", "|---|---|---|---|"]
    for r in rows[:n]:
        out.append("| `%s` | %s | %.1f | %s%% |" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1000, r['Percentage']))
    return "\n".join(out)


def last(f):
    return json.loads(open(root + f).read().strip().splitlines()[-1])


b, r, c = last('r02_bench.json'), last('r02_rtm_bench.json'), last('r02_c5_bench.json')
launches = [l for l in open(root + 'r02_step_timeline.txt').read().splitlines() if 'launches' in l][-1]
s = '''# Round 02 — summary of the committed measurements (MI355X, one GPU)

Everything below is produced by `bash tools/refresh_profiles_r02.sh` on the GPU box (then `python tools/install_profiles_r02.py r02`
and `python tools/make_profile_summary_r02.py`); the kernel statistics are `rocprofv3 --kernel-trace --stats --output-format csv --
python3 bench.py ...` summaries.

## C2 — `python bench.py` (BASELINE configs[1]: item_transformer d=128, bs 384, 20 negatives, dropout 0.1)

bench line (`r02_bench.json`): **%.0f tuples/s, %.4f ms/step**, median of 200 single steps %.4f ms (p10-p90 %.4f-%.4f), 13 kernel
launches per step (+ the stream write / wait-value operations of the side stream; `r02_step_timeline.txt`: %s).
Roofline object: `mlp_fwd_ws_kernel`, bound `mfma`, %.1f TFLOP/s of 157.3 = **%.3f** (in-step HIP-event duration %.1f µs; rocprof
average below); PMC utilisation of the matrix pipe `r02_mfma_utilisation.md`, instruction mix `r02_inst_counters.txt`, wait / busy
cycles `r02_sq_counters.txt`; why ~0.5 is the ceiling of that kernel: `r02_mlp_notes.md`.  Kernel statistics
(`r02_bench_kernel_stats.csv`, the timed steps plus the roofline pass):

%s

Round 1 for comparison (`r01_bench.json`): 22.3 M tuples/s, 0.344 ms/step, 18 launches.

## C4 — `python bench.py --workload c4` (BASELINE configs[3]: review_transformer, bs 256, K 5, R 20+30, WL 100, pvc)

bench line (`r02_rtm_bench.json`): **%.0f tuples/s, %.4f ms/step** (median %.4f), 20 launches (round 1: 0.733 ms, 27 launches).
Roofline object: `rtm_embed4_kernel`, bound `hbm`, %.0f GB/s of 8000 = **%.3f** on %.1f MB of algorithmic bytes (in-step %.1f µs):
the kernel is four dependent round trips per wave, not bytes (DESIGN.md 7c); PMC FETCH_SIZE / WRITE_SIZE passes `r02_rtm_embed_pmc.txt`
(the write counter includes the 1.16 M returning counter atomics).  Timeline `r02_rtm_step_timeline.txt`; kernel statistics
(`r02_rtm_kernel_stats.csv`):

%s

## C5 shard — `python bench.py --workload c5 --items 8000000` (one GPU's share of BASELINE configs[4]: d=256, bs 1024, row-sparse Adam)

bench line (`r02_c5_bench.json`): **%.0f tuples/s, %.3f ms/step**; roofline object: the stand-alone gather+score launch inside the step,
%.0f GB/s = %.3f of peak (it shares the machine there); alone on the chip at this shape: `r02_gather_c5_shape.jsonl` (0.61 of 8 TB/s at
B=1024, 0.71 at B=8192); PMC traffic `r02_gather_score_c5_pmc.txt` (63 MB against 67.6 MB algorithmic: no re-reads).
The d = 256 linears run as bf16x3 products (`gemm_x3_kernel`, DESIGN.md 5b): `r02_c5_bench_fp32_products.json` is the same run with
`PS_GEMM_X3=0` (1.578 ms/step); timeline `r02_c5_step_timeline.txt`, kernel statistics `r02_c5_kernel_stats.csv`; the form alone:
`r02_gemm_x3_bench.txt` (1.3-1.5x the fp32 MFMA kernel at equal error against fp64), counters `r02_gemm_x3_pmc.txt`.

## Late additions of the round (DESIGN.md 5, 5b, 7c)

bf16x3 product form of the GEMM kernel (exact three-way bf16 split, six bf16 MFMAs per product step, fp32-MFMA accuracy): d = 256
linears, 78k-row products, every flat weight-gradient group — `r02_gemm_x3_bench.txt`, `r02_gemm_x3_pmc.txt`, C5 1.58 -> 1.41 ms;
forks of the side stream signalled by the NEXT main-stream kernel (no stream operation on the main stream: the two bubbles of 9.5 and
10.5 us around the forks of the C2 backward are gone in `r02_step_timeline.txt`), value crossings on long steps too, the backward
tails rebalanced (K/V/Q weight gradients behind the scatter on the main stream; the review transformer's word-gradient reduce on the
side stream beside the query scatter): C2 0.285 -> 0.276, review transformer 0.529 -> 0.502, C5 shard 1.41 -> 1.36 ms per step.
Dropped on the way: two slabs of register prefetch in the bf16x3 kernel (same step time, 5 %% slower alone), a wide transposed fetch
for its weight gradients (mixed), the score scatter beside the fused backward (starved: 74 us instead of 29), a four-rows-per-trip
LayerNorm backward (no gain beside the side stream's atomics).

## Measured and dropped this round (numbers in DESIGN.md 5, 7c and the kernels' comments)

bf16x3 products in the fused forward (same accuracy, 54.7 vs 59.4 µs kernel, step unchanged: opt-in `PS_MLP_X3=1`); a two-level loss ticket;
one grouped weight-gradient launch on a max-shape grid (81 %% idle workgroups: 0.349 ms — the flat form fixed it); CU-masked side stream
(every kernel ~2x slower); lowest stream priority for the side stream (no effect, kept); LDS hash aggregation of the word counts
(slower: a chunk's words are mostly distinct); the inverted index built under the forward gather (slows it 87 -> 144 µs) or with the old
per-occurrence atomics; XCD column split of the word-gradient reduce (206 vs 66 µs); one wave per (sequence, head group) with 21 serial
replicas (37 µs) and eight waves per sequence (two rounds of 256-register workgroups, 40 µs in the step) before the four-wave form
(12.5 µs alone); a ticketed two-pass split reduction of the weight gradients (partials stored, last arriver adds them in split order:
deterministic, no fp32 atomics, but 122 vs 44 µs — the device-scope release before each workgroup's ticket writes back its XCD's L2; the
same fence doubled the review transformer's score kernel until its loss went through one fixed-point atomic instead).
''' % (b['value'], b['ms_per_step'], b['median_ms_per_step'], b['p10_p90_ms_per_step'][0], b['p10_p90_ms_per_step'][1], launches.strip(),
       b['roofline']['achieved'], b['roofline']['frac'], b['roofline']['us_per_launch'], table('r02_bench_kernel_stats.csv', 17),
       r['value'], r['ms_per_step'], r['median_ms_per_step'], r['roofline']['achieved'], r['roofline']['frac'],
       r['roofline']['bytes_per_launch'] / 1e6, r['roofline']['us_per_launch'], table('r02_rtm_kernel_stats.csv', 18),
       c['value'], c['ms_per_step'], c['roofline']['achieved'], c['roofline']['frac'])
open(root + 'r02_summary.md', 'w').write(s)
print(s[:900])
