#!/bin/bash
# Round-3 profile set for the workloads given (on the GPU box): bench line + rocprofv3 kernel stats + PMC summaries
#   tools/profile_sel_r03.sh S1 S2 M32
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for WL in "$@"; do
  python3 bench.py --workload $WL --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_bench_$WL.log 2> gpurun_out/r03_bench_$WL.err
  tools/profile_r03.sh $WL $WL > gpurun_out/r03_prof_$WL.log 2>&1
  echo "== $WL"; tail -1 gpurun_out/r03_bench_$WL.log | cut -c1-330
done
