# gpurun -- bash tools/env_rest_r05.sh : PMC traffic of the c2 / c4 steps, then the env-matrix configurations the first call did not reach
bash tools/c2_pmc_traffic.sh > /dev/null 2>&1
bash tools/pmc_traffic.sh c4 --workload c4 > /dev/null 2>&1
run() {
  echo "== $1" | tee -a gpurun_out/env_matrix_rest.log
  env $1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_rtm_shapes.py -q -x 2>&1 | tail -1 | tee -a gpurun_out/env_matrix_rest.log
}
: > gpurun_out/env_matrix_rest.log
for e in "PS_KVDX_FUSED=0" "PS_KVQ_FUSED=0 PS_KVDX_FUSED=0" "PS_SCORE_SIDX=0"; do run "$e"; done
for e in "PS_RTM_LATE_INDEX=1" "PS_RTM_EMBED4=0" "PS_WGRAD_GROUP_ROWS=0 PS_WG3_SIDE=0" "PS_FORK_BY_KERNEL=0" "PS_WG3_LAST=0" "PS_WG3_SIDE=1" "PS_SIDE_LIGHT=0" "PS_RTM_WR_SIDE=0" "PS_RTM_SBWD_SIG=0" "PS_X3_FLAT_SHAPE=1" "PS_X3_FLAT_SHAPE=2"; do
  run "PS_DIAG_LIB=1 $e"
done
