#!/bin/bash
# Round-4 evidence run (through gpurun from the repo root): bench lines of every workload, task-model steps, then the
# rocprofv3 kernel trace + PMC passes of S1 / S2 / S3 / M32 (tools/profile_r04.sh). Everything lands in gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
if [ "${SKIP_BENCH:-0}" = "0" ]; then
for w in S1 S2 S3 M32 H28 H16; do
  python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err
  echo "bench $w done: $(tail -c 300 gpurun_out/r04_bench_$w.json | head -c 120)"
done
python3 bench.py --deterministic --no-cpu-baseline > gpurun_out/r04_bench_S1_deterministic.json 2>/dev/null
for m in hulls md17 motion; do
  python3 tools/model_step_bench.py --model $m > gpurun_out/r04_${m}_step.log 2>&1
  tail -1 gpurun_out/r04_${m}_step.log
done
fi
# PART=1: the above + S1 / S2; PART=2: S3 / M32 / H28 + the task models' kernel statistics (two gpurun calls of <= 20 min)
for w in ${PROFILE_WORKLOADS:-S1 S2 S3 M32 H28}; do
  bash tools/profile_r04.sh $w $w > gpurun_out/r04_prof_$w.log 2>&1
  echo "profile $w done"
done
for m in ${PROFILE_MODELS:-md17 hulls}; do
  MODEL_ARGS="--model $m" bash tools/prof_model.sh > gpurun_out/r04_prof_model_$m.log 2>&1
  cp gpurun_out/hulls_model_kernel_stats.csv gpurun_out/r04_${m}_model_kernel_stats.csv
  echo "model profile $m done"
done
