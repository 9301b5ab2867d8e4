set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc1; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity_rtm.py tests/test_gpu_dp.py tests/test_gpu_rowsparse.py tests/test_gpu_parity.py -q -s > $O/tests.log 2>&1 || true
tail -15 $O/tests.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-extras > $O/a.log 2>&1
python tools/pmc_summary.py $O/a > $O/a.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-extras > $O/b.log 2>&1
python tools/pmc_summary.py $O/b > $O/b.txt
rm -rf $O/a $O/b
cat $O/a.txt | cut -c1-220
