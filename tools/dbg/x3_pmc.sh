set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/x3pmc; rm -rf $O; mkdir -p $O
ARGS="21504 1024 256 0 0 0"
for form in "1 1" "0 -1"; do
  tag=$(echo $form | tr ' -' '__')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq_$tag -- python3 tools/gemm_x3_one.py $ARGS $form > $O/sq_$tag.log 2>&1
  python tools/pmc_summary.py $O/sq_$tag | grep -E "kernel|gemm" | cut -c1-220 > $O/sq_$tag.txt
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/in_$tag -- python3 tools/gemm_x3_one.py $ARGS $form > $O/in_$tag.log 2>&1
  python tools/pmc_summary.py $O/in_$tag | grep -E "kernel|gemm" | cut -c1-220 > $O/in_$tag.txt
  rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/mem_$tag -- python3 tools/gemm_x3_one.py $ARGS $form > $O/mem_$tag.log 2>&1 || true
  python tools/pmc_summary.py $O/mem_$tag | grep -E "kernel|gemm" | cut -c1-220 > $O/mem_$tag.txt || true
  find $O -name '*agent_info.csv' -delete
done
cat $O/*.txt
