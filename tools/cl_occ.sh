#!/bin/bash
# occupancy experiment: forward / backward kernel time against resident workgroups per CU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for capf in 256 512 768 1024 1280; do
  export CSMPN_CL_CAP_FWD=$capf CSMPN_CL_CAP_BWD=$(( capf > 512 ? 512 : capf ))
  rm -rf gpurun_out/occ_$capf
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/occ_$capf -- python3 tools/cl_stage4.py S1 > /dev/null 2>&1
  echo "cap fwd $CSMPN_CL_CAP_FWD bwd $CSMPN_CL_CAP_BWD"; python3 tools/kstats.py gpurun_out/occ_$capf | grep "cemlp_cl.*<8, 1"
done
