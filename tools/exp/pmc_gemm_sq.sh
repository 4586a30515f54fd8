set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm_sq
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p3.log 2>&1
ls $O
