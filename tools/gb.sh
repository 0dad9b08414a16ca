#!/bin/bash
# Build the library, then run bench.py (S1, graph replay, no CPU baseline) on the GPU box and print the stage times.
#   tools/gb.sh <tag> [extra bench args]
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift
make -C clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc -j8 2>&1 | grep -E "error|Error" && exit 1
/usr/local/graft/bin/gpurun --timeout 300 -- "python bench.py --steps 100 --warmup 10 --no-cpu-baseline $* > gpurun_out/r03_gb_$TAG.log 2>gpurun_out/r03_gb_$TAG.err; tail -3 gpurun_out/r03_gb_$TAG.err" 2>&1 | grep -v "^\[gpurun\] \(sending\|merged\)" | tail -4
python3 - <<PY
import json
r=json.loads(open("gpurun_out/r03_gb_$TAG.log").read().strip().split("\n")[-1])
print("$TAG", "value %.4g edges/s" % r["value"], "ms/step", r["ms_per_step"], r["roofline"]["stage_ms"])
PY
