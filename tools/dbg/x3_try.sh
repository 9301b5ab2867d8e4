set -e
O=gpurun_out/x3; mkdir -p $O
for s in 0 1 2; do
  PS_GEMM_X3=1 PS_GEMM_X3_SHAPE=$s timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "mfma_gemm" > $O/parity_s$s.log 2>&1 || { tail -30 $O/parity_s$s.log; exit 1; }
  tail -1 $O/parity_s$s.log
done
timeout -k 10 600 python tools/gemm_x3_bench.py > $O/bench.log 2>&1 || { tail -30 $O/bench.log; exit 1; }
cat $O/bench.log
for ks in 8 16 32; do
  PS_GEMM_KSPLIT=$ks timeout -k 10 300 python tools/gemm_x3_bench.py wgrad > $O/wgrad_$ks.log 2>&1 || { tail -30 $O/wgrad_$ks.log; exit 1; }
  cat $O/wgrad_$ks.log
done
