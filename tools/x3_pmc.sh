# gpurun -- bash tools/x3_pmc.sh : wave-state and LDS counters of the three bf16x3 GEMM kernels on one product (counter passes alone)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/x3pmc; rm -rf $O; mkdir -p $O
M=${M:-4096}; N=${N:-4096}; K=${K:-4096}
for shp in ${SHAPES:-2 3 4}; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq$shp -- python3 tools/gemm_x3_one.py $M $N $K 0 0 0 1 $shp 6 > $O/sq$shp.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/in$shp -- python3 tools/gemm_x3_one.py $M $N $K 0 0 0 1 $shp 6 > $O/in$shp.log 2>&1
  echo "== shape $shp"; python tools/pmc_summary.py $O/sq$shp | grep -E "kernel|gemm_x3" | cut -c1-220; python tools/pmc_summary.py $O/in$shp | grep -E "kernel|gemm_x3" | cut -c1-220
done
rm -rf $O/sq? $O/in?
