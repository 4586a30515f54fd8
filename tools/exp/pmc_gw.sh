set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gw
rm -rf $O; mkdir -p $O
V=${1:-0}
run() { # name counters...
  n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pgw_$n -o p -- python3 $R/tools/exp/gw_one.py 4 $V > $O/$n.log 2>&1
  python3 $R/tools/rocprof_pmc_summary.py $(find /tmp/pgw_$n -name '*.db' | head -1) $O/$n.csv > /dev/null
  rm -rf /tmp/pgw_$n
}
run lds SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run mfma SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
grep -h gemm_wide $O/*.csv
