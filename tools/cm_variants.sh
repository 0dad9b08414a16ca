#!/bin/bash
# Timing experiments on the channel-MFMA kernels: one library per switch (results WRONG, timing only; built into
# tools/_bin/, removed after the run). tools/cm_variants.sh NOATOM NOSAVE ... ; then run tools/cl_variants_run.sh S2 on the box.
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
B=_build
for V in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -DCM_X_$V -c k_cm_n3.hip -o $B/k_cm_n3_x.o
  hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/csr.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/glue.o $B/layers.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o $B/k_cl_n3.o $B/k_cm_n3_x.o -o ../../tools/_bin/libx_$V.so
  echo built $V
done
