#!/bin/bash
# On the GPU box: per-kernel averages (rocprofv3 kernel trace) of the shipped library and of every tools/_bin/libx_*.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
WL=${1:-S1}
run() {  # tag, lib ('' = shipped)
  rm -rf gpurun_out/xv_$1
  if [ -n "$2" ]; then export CSMPN_LIB=$2; else unset CSMPN_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xv_$1 -- python3 tools/cl_stage4.py $WL > gpurun_out/xv_$1.log 2>&1
  python3 tools/kstats.py gpurun_out/xv_$1 | grep -v "Fill\|copyBuffer"
}
run base ""
for f in tools/_bin/libx_*.so; do
  [ -e "$f" ] || continue
  t=$(basename $f .so); run ${t#libx_} $PWD/$f
done
