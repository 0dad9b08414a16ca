#!/bin/bash
# Iteration aid: rebuild only the Cl(3,0) unit (and capi.hip when asked) and relink with the
# other, already built objects. Use `make` for a full, consistent build before committing.
set -e
cd "$(dirname "$0")/../clifford-group-equivariant-simplicial-message-passing-networks_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value"
B=${BUILD:-_build}
OUT=${OUT:-../csmpn_hip/libcsmpn_hip.so}
hipcc $FLAGS $EXTRA -mllvm -amdgpu-mfma-vgpr-form -c k_n3.hip -o $B/k_n3.o &
if [ "$1" = "capi" ]; then hipcc $FLAGS $EXTRA -c capi.hip -o $B/capi.o & fi
wait
hipcc -shared -fPIC --offload-arch=gfx950 $B/capi.o $B/k_n2.o $B/k_n3.o $B/k_n4.o $B/k_n4m.o $B/k_n5.o $B/k_n5m.o -o $OUT
echo built $OUT
