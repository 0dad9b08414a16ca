#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_rl2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/rl_check.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VSKIPPED SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 tools/rl_check.py > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    files = glob.glob(f"gpurun_out/pmc_rl2/{p}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            if "rl_kernel" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(p, k)
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v / cnt[(k, c)]:16.0f}")
PY
tail -3 $OUT/p1.log $OUT/p2.log
