for e in "PS_FORK_BY_KERNEL=0" "PS_WG3_LAST=0" "PS_WG3_SIDE=1" "PS_SIDE_LIGHT=0"; do
echo "== $e" | tee -a gpurun_out/fork_matrix.log
env $e timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_rtm_shapes.py -q -x 2>&1 | tail -1 | tee -a gpurun_out/fork_matrix.log
done
