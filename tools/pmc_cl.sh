#!/bin/bash
# PMC passes (no trace domains) over tools/cl_stage4.py; per-kernel averages -> gpurun_out/r03_<tag>_pmcx.json
#   tools/pmc_cl.sh <tag> [workload = S1] [kernel-name filter = cemlp_cl]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=${1:-x}; WL=${2:-S1}; FILTER=${3:-cemlp_cl}
OUT=gpurun_out/pmcx_$TAG
rm -rf $OUT; mkdir -p $OUT
P="python3 tools/cl_stage4.py $WL"
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $OUT/pmc1 -- $P > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc2 -- $P > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc3 -- $P > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES --output-format csv -d $OUT/pmc4 -- $P > $OUT/p4.log 2>&1
python3 - <<PY
import collections, csv, glob, json, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "$FILTER" not in k: continue
        k = k.replace("csmpn::Alg<3, 0u>, ", "").replace("(csmpn::DevCemlp, csmpn::RowIO)", "").replace("void csmpn::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
summary = {k: {c: int(round(v / cnt[k][c])) for c, v in sorted(d.items())} for k, d in acc.items()}
json.dump(summary, open("gpurun_out/r03_${TAG}_pmcx.json", "w"), indent=1)
for k, d in summary.items():
    print(k)
    print("  ", {c: v for c, v in d.items()})
PY
