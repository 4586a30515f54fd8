# Collect the rocprofv3 evidence of all four bench workloads (kernel trace + FETCH / WRITE / SQ passes each) under TAG=r3:
#     bash tools/collect_round.sh   (on the GPU box; results under gpurun_out/)
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for w in student sde tiled1080 dataset512; do
  bash tools/collect_profiles.sh r3_$w --workload $w --no-overlap --serial-branches
  grep '^{' gpurun_out/r3_${w}_ks.log | tail -1 > gpurun_out/r3_${w}_bench_line_under_trace.json
  echo done $w
done
