set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc2; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-extras > $O/a.log 2>&1
python tools/pmc_summary.py $O/a | grep -E "kernel|mlp_" > $O/a.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-extras > $O/b.log 2>&1
python tools/pmc_summary.py $O/b | grep -E "kernel|mlp_" > $O/b.txt
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $O/c -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-extras > $O/c.log 2>&1 || true
python tools/pmc_summary.py $O/c | grep -E "kernel|mlp_" > $O/c.txt || true
rm -rf $O/a $O/b $O/c
cut -c1-230 $O/a.txt; cut -c1-230 $O/b.txt; cut -c1-230 $O/c.txt
