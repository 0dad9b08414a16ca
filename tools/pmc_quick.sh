#!/bin/bash
# quick PMC pass over bench.py (kernel-level instruction counts); usage: tools/pmc_quick.sh <tag> [env...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY} --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name']
        if 'cemlp' not in k: continue
        k=k.split('(')[0][-60:]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k in acc:
    d=acc[k]; m=max(n[k],1)
    print(k, 'launches',m, {c:round(v/m) for c,v in d.items()})
PY
