#!/bin/bash
# Diagnostic library with per-phase s_memtime stamps in the (row, channel)-per-lane kernels (never shipped, never
# timed): tools/_bin/libcsmpn_hip_stamps.so, read by tools/cl_stamps.py. Remove it after use (it travels with gpurun).
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
B=_build
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -fno-slp-vectorize -DCSMPN_STAMPS -c k_cl_n3.hip -o $B/k_cl_n3_stamps.o
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/csr.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o $B/glue.o $B/layers.o $B/k_pl_n5.o $B/k_pl_n5m.o $B/k_plw_n5.o $B/k_plw_n5m.o $B/k_cl_n3_stamps.o -o ../../tools/_bin/libcsmpn_hip_stamps.so
echo built tools/_bin/libcsmpn_hip_stamps.so
