#!/bin/bash
# Round-4 evidence run (through gpurun from the repo root): bench lines of every workload, task-model steps, then the
# rocprofv3 kernel trace + PMC passes of S1 / S2 / S3 / M32 (tools/profile_r04.sh). Everything lands in gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for w in S1 S2 S3 M32 H28 H16; do
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err
  echo "bench $w done: $(tail -c 300 gpurun_out/r04_bench_$w.json | head -c 120)"
done
python3 bench.py --deterministic --no-cpu-baseline > gpurun_out/r04_bench_S1_deterministic.json 2>/dev/null
for m in hulls md17 motion; do
  python3 tools/model_step_bench.py --model $m > gpurun_out/r04_${m}_step.log 2>&1
  tail -1 gpurun_out/r04_${m}_step.log
done
for w in S1 S2 S3 M32; do
  bash tools/profile_r04.sh $w $w > gpurun_out/r04_prof_$w.log 2>&1
  echo "profile $w done"
done
