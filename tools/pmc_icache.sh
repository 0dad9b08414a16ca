#!/bin/bash
# Instruction-cache counters of the S1 kernels (separate --pmc pass, no trace domains).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python3 bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 2 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_icache/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_IFETCH': n[k] += 1
for k, v in acc.items():
    if 'cemlp_rl' in k:
        print(k, 'launches', n[k]); print('   ', {a: round(b / max(n[k], 1)) for a, b in v.items()})
PY
