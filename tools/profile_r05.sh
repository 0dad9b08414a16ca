#!/bin/bash
# Round-5 profiling recipe (run on the GPU box through gpurun from the repo root):
#   tools/profile_r05.sh <tag> [workload] [quick]
# Kernel trace + stats first, PMC counters in their own passes (no trace domains), then
# tools/summarize_profile.py writes r05_<tag>_kernel_stats.csv and r05_<tag>_pmc_summary.json into gpurun_out/
# for copying into profiles/. `quick` = the kernel trace only.
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
WL=${2:-S1}
B="python3 bench.py --no-cpu-baseline --no-graph --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
if [ "$3" != "quick" ]; then
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- $B --steps 3 --warmup 2 > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc2 -- $B --steps 3 --warmup 2 > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $B --steps 3 --warmup 2 > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc4 -- $B --steps 3 --warmup 2 > $OUT/bench_pmc4.log 2>&1
fi
python3 tools/summarize_profile.py $OUT $TAG r05
