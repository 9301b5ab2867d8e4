# gpurun -- bash tools/gather_variants.sh : the stand-alone gather+score launch at the C5 shape (8 M-row table, rotating index
# sets) under the diagnostic library's launch-shape knobs
export PS_DIAG_LIB=1
out=gpurun_out/r04_gather_variants.txt
: > $out
for e in "PS_SCORE_CH=4" "PS_SCORE_CH=8" "PS_SCORE_CH=2" "PS_SCORE_CH=4 PS_SCORE_WIDE_U=2" "PS_SCORE_CH=8 PS_SCORE_WIDE_U=2"; do
  for b in 1024 8192; do
    env $e timeout -k 10 120 python tools/gather_c5.py --rows 8000000 --batch $b --iters 48 2>/dev/null | tail -1 >> $out || exit 1
  done
done
env PS_SCORE_CH=4 timeout -k 10 120 python tools/gather_c5.py --rows 8000000 --batch 1024 --iters 48 --sets 1 2>/dev/null | tail -1 >> $out
cut -c1-60,150-400 $out
