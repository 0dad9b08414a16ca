#!/bin/bash
# PMC counters of the row-per-lane kernels on the S1 shape (run on the GPU box through gpurun).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_rl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 tools/rl_check.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 tools/rl_check.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p3 -- python3 tools/rl_check.py > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    files = glob.glob(f"gpurun_out/pmc_rl/{p}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            if "rl_kernel" not in k and "reduce" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(p, k)
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v / cnt[(k, c)]:16.0f}  (per launch, {cnt[(k,c)]} launches)")
PY
