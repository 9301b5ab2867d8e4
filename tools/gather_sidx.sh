# gpurun -- bash tools/gather_sidx.sh : the stand-alone gather+score launch at the C5 shape (8 M-row table, rotating index sets),
# the index hop on the scalar path (default, round 5) against the per-lane form (PS_SCORE_SIDX=0); then the per-workgroup timelines
out=gpurun_out/r05_gather_sidx.txt
: > $out
for e in "PS_SCORE_SIDX=1" "PS_SCORE_SIDX=0" "PS_SCORE_SIDX=1" "PS_SCORE_SIDX=0"; do
  for b in 1024 8192; do
    env $e timeout -k 10 120 python tools/gather_c5.py --rows 8000000 --batch $b --iters 48 2>/dev/null | tail -1 >> $out || exit 1
  done
done
export PS_DIAG_LIB=1
for e in "PS_SCORE_CH=4" "PS_SCORE_CH=8"; do
  env $e timeout -k 10 120 python tools/gather_c5.py --rows 8000000 --batch 1024 --iters 48 2>/dev/null | tail -1 >> $out || exit 1
  env $e timeout -k 10 120 python tools/gather_c5.py --rows 8000000 --batch 8192 --iters 48 2>/dev/null | tail -1 >> $out || exit 1
done
cut -c1-60,150-420 $out
w=gpurun_out/r05_gather_score_wg_times.txt
: > $w
env PS_SCORE_CH=4 timeout -k 10 120 python tools/gather_wg_times.py --batch 1024 >> $w 2>&1 || exit 1
env PS_SCORE_CH=8 timeout -k 10 120 python tools/gather_wg_times.py --batch 1024 >> $w 2>&1 || exit 1
env PS_SCORE_CH=4 PS_SCORE_SIDX=0 timeout -k 10 120 python tools/gather_wg_times.py --batch 1024 >> $w 2>&1 || exit 1
cat $w
