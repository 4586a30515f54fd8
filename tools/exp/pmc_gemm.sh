set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/p3 -o p3 --output-format csv -- python3 $R/tools/exp/gemm_one.py 5 > $O/p3.log 2>&1
ls -R $O | head -30
