#!/bin/bash
# Collect the round's rocprofv3 evidence for `bench.py` on the GPU box: kernel-trace stats, then (unless PMC=0) one PMC
# pass per counter group (FETCH_SIZE, WRITE_SIZE, SQ: separate runs, as MI355X_MICROARCH.md prescribes), each summarised
# to a small CSV under gpurun_out/ (the rocpd databases are too large to carry back).
#     bash tools/collect_profiles.sh TAG [bench.py arguments, e.g. --workload sde]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r2}
shift || true
ARGS="$@"
KS_STEPS=${KS_STEPS:-4}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/prof_ks /tmp/prof_pmc
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python3 $R/bench.py --steps $KS_STEPS --warmup 2 --cpu-baseline skip --census skip $ARGS > $OUT/${TAG}_ks.log 2>&1
python3 $R/tools/rocprof_db_to_csv.py $(find /tmp/prof_ks -name '*.db' | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf /tmp/prof_ks
if [ "${PMC:-1}" != "0" ]; then
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    timeout -s KILL 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/prof_pmc -o pm -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-baseline skip --census skip $ARGS > $OUT/${TAG}_pmc_${name}.log 2>&1
    python3 $R/tools/rocprof_pmc_summary.py $(find /tmp/prof_pmc -name '*.db' | head -1) $OUT/${TAG}_pmc_${name}.csv
    rm -rf /tmp/prof_pmc
done
fi
echo collected $TAG
