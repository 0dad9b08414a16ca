#!/bin/bash
# Round-3 profile set (on the GPU box): bench lines + rocprofv3 kernel stats + PMC summaries of S1, S2, S3, M32
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for WL in S1 S2 S3 M32; do
  python3 bench.py --workload $WL --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_bench_$WL.log 2> gpurun_out/r03_bench_$WL.err
  tools/profile_r03.sh $WL $WL > gpurun_out/r03_prof_$WL.log 2>&1
  echo "== $WL"; tail -1 gpurun_out/r03_bench_$WL.log | cut -c1-330
done
