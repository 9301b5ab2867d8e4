for e in "PS_GEMM_X3_SHAPE=0" "PS_GEMM_X3_SHAPE=1" "PS_GEMM_X3_SHAPE=2" "PS_GEMM_X3=0"; do
echo "== $e" | tee -a gpurun_out/x3_matrix.log
env $e timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_rtm_shapes.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -4 | tee -a gpurun_out/x3_matrix.log
done
