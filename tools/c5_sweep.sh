# Tuning sweep of the gather+score launch (profiles/r01_gather_score_tuning.txt): chunks per lane (PS_SCORE_CH: 0 = the
# one-chunk form, 2 / 4 / 8) and tasks per row group at the C5 shape, then the C2 launch through bench.py.  Run on the GPU box.
for W in 0 1; do
for B in 1024 8192; do
  for cfg in "PS_SCORE_CH=0" "PS_SCORE_CH=2" "PS_SCORE_CH=4" "PS_SCORE_CH=4 PS_SCORE_WIDE_U=2" "PS_SCORE_CH=8"; do
    echo "W=$W B=$B $cfg: $(env $cfg python tools/gather_c5.py --rows 8000000 --batch $B --iters 50 --w $W 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f us %.0f GB/s" % (j["us_per_launch"], j["achieved_GBps"]))')"
  done
done
done
for cfg in "PS_SCORE_CH=0" "PS_SCORE_CH=2" "PS_SCORE_CH=4"; do
  echo "C2 $cfg: $(env $cfg python bench.py --cpu-steps 0 --steps 200 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f ms/step, gather+score %.2f us %.0f GB/s" % (j["ms_per_step"], j["roofline"]["us_per_launch"], j["roofline"]["achieved"]))')"
done
