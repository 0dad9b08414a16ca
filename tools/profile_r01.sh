#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun from the repo root).
# Kernel trace + stats first, PMC counters in their own passes (no trace domains).
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc4 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
