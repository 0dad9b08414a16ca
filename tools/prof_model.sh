#!/bin/bash
# rocprofv3 kernel stats of the whole-model hulls training step (tools/model_step_bench.py); run through gpurun.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_hulls_model
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/model_step_bench.py --steps 10 $MODEL_ARGS > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/hulls_model_kernel_stats.csv", "w") as o:
    o.write(open(f).read())
for r in rows[:16]:
    print(r["Name"][:120], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 $OUT/run.log
