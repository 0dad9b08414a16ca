#!/bin/bash
# Diagnostic library with per-phase s_memtime stamps in the Cl(3,0) 16-row-tile MFMA-mixing kernels (cemlp_pq.hpp; never shipped,
# never timed): tools/_bin/libcsmpn_hip_stamps.so, read by tools/pg_stamps.py M32. Remove it after use (it travels with gpurun).
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
B=_build
mkdir -p ../../tools/_bin
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value"
hipcc $F -fno-slp-vectorize -DCSMPN_STAMPS -c k_pq_n3.hip -o $B/k_pq_n3_stamps.o &
[ $B/capi_stamps.o -nt capi.hip ] || hipcc $F -DCSMPN_STAMPS -c capi.hip -o $B/capi_stamps.o &
wait
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi_stamps.o $B/csr.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/glue.o $B/layers.o $B/k_cl_n3.o $B/k_cm_n3.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o $B/k_pg_n5.o $B/k_pg_n5m.o $B/k_pq_n3_stamps.o -o ../../tools/_bin/libcsmpn_hip_stamps.so
echo built tools/_bin/libcsmpn_hip_stamps.so
