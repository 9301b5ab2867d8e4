# Runs the parity suites under every alternative code-path switch on the GPU box:
#   gpurun -- bash tools/env_matrix.sh supported   and   gpurun -- bash tools/env_matrix.sh diag
#   (25 + 11 configurations of ~42 s: the two halves are 17 and 8 minutes)
# First the SUPPORTED switches against the shipped library, then the diagnostic build's scheduling / launch-shape knobs that select
# whole alternative paths (PS_DIAG_LIB=1: python -m prodsearch_amd.build --diag must have run).
run() {
  echo "== $1" | tee -a gpurun_out/env_matrix.log
  env $1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_rtm.py tests/test_gpu_shapes.py tests/test_gpu_rtm_shapes.py -q -x 2>&1 | tail -1 | tee -a gpurun_out/env_matrix.log
}
part=${1:-all}
[ "$part" = diag ] || : > gpurun_out/env_matrix.log
[ "$part" = diag ] || for e in "PS_NO_SIDE=1" "PS_SIDE_EVENTS=1" "PS_NO_FUSE=1" "PS_NO_FUSE_BWD=1" "PS_ATTN_WF=0 PS_ATTN_W1=0" "PS_NO_ROWLIST=1" "PS_KEEP_GRADS=1" "PS_NO_FOLD_SCORE=1" "PS_DETERMINISTIC=1" "PS_GEMM_X3=0" "PS_GEMM_X3_SHAPE=0" "PS_GEMM_X3_SHAPE=1" "PS_GEMM_X3_SHAPE=2" "PS_GEMM_X3_SHAPE=3" "PS_GEMM_X3_SHAPE=4" "PS_RTM_HIST=0" "PS_RTM_GROUPLIST=0" "PS_DP_RS=a2a PS_DP_AG=a2a" "PS_GRAPHS=1" "PS_KVQ_FUSED=0" "PS_KVDX_FUSED=0" "PS_KVQ_FUSED=0 PS_KVDX_FUSED=0" "PS_SCORE_SIDX=0" "PS_FW_BY_GEMM=0" "PS_ATTN_WK=0"; do
  run "$e"
done
[ "$part" = supported ] || for e in "PS_RTM_LATE_INDEX=1" "PS_RTM_EMBED4=0" "PS_WGRAD_GROUP_ROWS=0 PS_WG3_SIDE=0" "PS_FORK_BY_KERNEL=0" "PS_WG3_LAST=0" "PS_WG3_SIDE=1" "PS_SIDE_LIGHT=0" "PS_RTM_WR_SIDE=0" "PS_RTM_SBWD_SIG=0" "PS_X3_FLAT_SHAPE=1" "PS_X3_FLAT_SHAPE=2"; do
  run "PS_DIAG_LIB=1 $e"
done
