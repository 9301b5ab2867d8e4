"""profiles/<round>_mfma_utilisation.md from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE run of bench.py.
    python tools/mfma_util.py <counter dir> <round tag> > profiles/<round>_mfma_utilisation.md"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))[0]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r01'
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = '`%s | grid %s`' % (r['Kernel_Name'][:52], r.get('Grid_Size', '?'))
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); cnt[k] += 1
print("# Round %s — MFMA utilisation per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, MI355X)\n" % rnd[1:])
print("Command: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 20 "
      "--warmup 5 --cpu-steps 0 --no-extras`\n")
print("`SQ_VALU_MFMA_BUSY_CYCLES` = 64 cycles per `v_mfma_f32_32x32x2_f32` summed over the chip; `GRBM_GUI_ACTIVE` is the sum over "
      "the 8 XCDs, so utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs).  fp32 MFMA peak = 1024 SIMDs x 4096 flop / 64 "
      "cycles x 2.4 GHz = 157 TFLOP/s.  Counter passes serialise dispatches; durations under `--pmc` are not the step's.\n")
print("| kernel | launches | MFMA busy cycles / launch | active cycles / XCD | MFMA utilisation |\n|---|---|---|---|---|")
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
    m = acc[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / cnt[k]; g = acc[k].get('GRBM_GUI_ACTIVE', 0) / cnt[k] / 8
    if m <= 0:
        continue
    print("| %s | %d | %.0f | %.0f | %.1f %% |" % (k, cnt[k], m, g, 100 * m / (g * 1024) if g else 0))
